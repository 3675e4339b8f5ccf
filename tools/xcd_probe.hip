// Where does the dispatcher put the workgroups of a launch?  Prints HW_REG_XCC_ID of the first 32 workgroups of three
// grids and whether "workgroup b sits on XCD (b + off) % 8" holds for the whole grid (and with which off).
//   hipcc --offload-arch=gfx950 -O2 tools/xcd_probe.hip -o tools/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;
}
int main() {
    const int grids[3] = {256, 1024, 2048};
    unsigned *d;
    hipMalloc(&d, 4096 * sizeof(unsigned));
    for (int rep = 0; rep < 2; ++rep)
        for (int g : grids) {
            hipLaunchKernelGGL(probe, dim3(g), dim3(256), 0, 0, d);
            std::vector<unsigned> x(g);
            hipMemcpy(x.data(), d, g * sizeof(unsigned), hipMemcpyDeviceToHost);
            int off = (int)((x[0] + 8 - 0) % 8), rot = 1, perm = 1;
            for (int b = 0; b < g; ++b) {
                if (x[b] != (unsigned)((b + off) % 8)) rot = 0;
                if (x[b] != x[b % 8]) perm = 0;
            }
            printf("grid %4d:", g);
            for (int b = 0; b < 32; ++b) printf(" %u", x[b]);
            printf(" | rotation by %d holds: %d | residue-consistent: %d\n", off, rot, perm);
        }
    return 0;
}
