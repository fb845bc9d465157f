class _Sub:
    def __getitem__(self, item): return item if not isinstance(item, tuple) else item[0]
    def __call__(self, *a, **k): return self
    def __getattr__(self, n): return _Sub()
conf = _Sub()
class _Extras:
    def subcommand_type_from_defaults(self, *a, **k): return object
    def __getattr__(self, n): return _Sub()
extras = _Extras()
def cli(*a, **k): raise RuntimeError("tyro stub")
