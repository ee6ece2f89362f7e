"""Importable alias of the product package directory `vln-goat_amd/` (a hyphen is not a valid Python
identifier, so `import vln_goat_amd` resolves here and re-targets the package path)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'vln-goat_amd')
__path__ = [_real]
with open(_os.path.join(_real, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))
