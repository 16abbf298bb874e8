def register(id=None, entry_point=None, **kwargs):
    return None
