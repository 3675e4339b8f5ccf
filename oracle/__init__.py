"""TEST INFRASTRUCTURE ONLY — the checker side of this repository: a CPU restatement of the reference's algorithms for
the hot path (cornac_oracle.c + the *_oracle.py files), the recipe that builds the real reference's extensions from
/root/reference where it exists (build_ref.py, ref_loader.py) and the stand-in modules they load over (ref_stubs/).
Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only; nothing under cornac_amd/ uses it."""
