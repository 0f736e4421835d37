"""Importable alias of the ``nasa-niswan_amd/`` package directory (a hyphen cannot appear in a
Python module name).  All code lives in ``nasa-niswan_amd/``; this file only redirects."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "nasa-niswan_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
