def is_bearable(obj, hint):
    """Only used by the reference to test for Tuple[str, ...] / List[str]."""
    return isinstance(obj, (tuple, list)) and all(isinstance(o, str) for o in obj)
