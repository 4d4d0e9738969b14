"""Config container with the reference's attribute-style access (utils.py:10-21)."""


class AttrDict(dict):
    """Dictionary whose keys can be read as attributes; nested dicts are wrapped on access."""

    def __init__(self, *args, **kwargs):
        super(AttrDict, self).__init__(*args, **kwargs)

    def __getattr__(self, item):
        try:
            v = self[item]
        except KeyError:
            raise AttributeError(item)
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def __setattr__(self, key, value):
        self[key] = value
