"""Stub: `beartype` as an identity decorator (also when called with keyword configuration)."""
def beartype(obj=None, **_kw):
    if obj is None:
        return lambda o: o
    return obj
