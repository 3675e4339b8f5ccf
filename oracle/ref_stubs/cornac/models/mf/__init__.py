"""stand-in package: lets oracle/ref_loader resolve the compiled cornac.models.mf.backend_cpu on the GPU box"""
