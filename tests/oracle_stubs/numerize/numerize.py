def numerize(n, *a, **k):
    return str(n)
