def save_image(*a, **k):
    return None
