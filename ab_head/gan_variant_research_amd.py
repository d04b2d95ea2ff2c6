"""Import shim: the package directory is named ``gan-variant-research_amd`` (not a Python identifier), so this module
turns itself into that package: ``import gan_variant_research_amd.cut`` loads ``gan-variant-research_amd/cut.py``."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "gan-variant-research_amd")]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
