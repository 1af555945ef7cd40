"""Import alias: the product package lives in the directory ``glow-tts_amd/`` (the name the
project layout prescribes), which is not a valid Python identifier.  This stub makes it
importable as ``glow_tts_amd`` by pointing the package search path at that directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "glow-tts_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
