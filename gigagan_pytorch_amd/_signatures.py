"""ctypes signatures of the element-wise / reduction entry points of include/gigagan_amd.h."""
import ctypes as C


def declare(L):
    pass
