"""stub package: the compiled extension modules of this package are resolved to oracle/_ref by the loader's finder"""
