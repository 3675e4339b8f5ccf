"""stub package: see ../../README.md"""
