"""TEST INFRASTRUCTURE: minimal stand-in for the third-party ``addict`` package (``Dict``), which the reference's
``ml3d/utils/config.py:9`` imports and this image does not ship.  Only tests that import the reference checkout put this
directory on the path; the product never does."""


class Dict(dict):

    def __init__(self, *args, **kwargs):
        super().__init__()
        for a in args:
            if a is None:
                continue
            for k, v in dict(a).items():
                self[k] = v
        for k, v in kwargs.items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, Dict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            return self.__missing__(name)

    def __missing__(self, name):
        return type(self)()

    def __setattr__(self, name, value):
        self[name] = value

    def __setitem__(self, name, value):
        super().__setitem__(name, self._wrap(value))

    def copy(self):
        return type(self)(self)

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Dict) else v) for k, v in self.items()}
