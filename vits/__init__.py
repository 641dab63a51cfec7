"""`vits` -- import-path shim so the reference's scripts (`from vits.light.vcvits import VCVITS`,
`import vits.commons`, `from vits.hparams import HParams`, ...) resolve to the MI355X-native
implementation in `vcvits_amd` without edits.  Every `vits.X.Y` is the SAME module object as
`vcvits_amd.X.Y` (no second copy is imported)."""
import importlib
import importlib.abc
import importlib.util
import sys

import vcvits_amd

_PREFIX = __name__ + "."
_REAL = "vcvits_amd."


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        real = _REAL + fullname[len(_PREFIX):]
        try:
            importlib.import_module(real)
        except ImportError:
            return None
        return importlib.util.spec_from_loader(fullname, self, is_package=hasattr(sys.modules[real], "__path__"))

    def create_module(self, spec):
        return sys.modules[_REAL + spec.name[len(_PREFIX):]]

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _AliasFinder())
__path__ = []  # all submodules come from the finder above
