import sys; sys.path.insert(0, "/root/repo")
from cornac_amd import _lib
for mb in (32, 64, 128, 192, 256, 384, 512, 1024, 2048, 6144):
    r = _lib.device_probe(0, mb << 20)
    print("%5d MiB: copy %.0f  stream %.0f  gather512 %.0f GB/s" % (mb, r["d2d_copy_GBps"], r["stream_read_GBps"], r["row_gather_512B_GBps"]), flush=True)
