__version__ = '0.3.0'   # tracks the reference checkpoint format version (gigagan_pytorch/version.py)
