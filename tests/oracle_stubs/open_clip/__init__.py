"""Stub: open_clip is never called on the unconditional path; constructing a real adapter must fail loudly."""
def create_model_and_transforms(*a, **k):
    raise RuntimeError('open_clip stub: no CLIP weights offline; pass pre-computed text_encodings')


def get_tokenizer(*a, **k):
    raise RuntimeError('open_clip stub: no tokenizer offline')
