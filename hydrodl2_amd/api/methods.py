"""Model discovery / loading with the reference's semantics.

Reference: src/hydrodl2/api/methods.py:18-144.  Models live in
`hydrodl2_amd/models/<family>/<snake_name>.py`, one class per file; the family is
the first word of the snake-case name.  `load_model(name, ver_name=None)`
returns the *class*; when no class called `ver_name` exists in the file, the
first class defined there is returned with a warning (same fallback as the
reference, restricted to classes the file itself defines so that imports such
as `HbvModule` are never picked).
"""
from __future__ import annotations

import importlib
import logging
import os
import re
from pathlib import Path

from torch.nn import Module

log = logging.getLogger("hydrodl2_amd")

_PKG = Path(os.path.dirname(os.path.abspath(__file__))).parent
_AVOID = {"__init__", ".DS_Store", "README", ".git"}


def _scan(sub: str) -> dict[str, list[str]]:
    root = _PKG / sub
    found: dict[str, list[str]] = {}
    if not root.is_dir():
        return found
    for d in sorted(root.iterdir()):
        if d.is_dir() and d.name != "__pycache__":
            names = [os.path.splitext(f.name)[0] for f in sorted(d.iterdir())
                     if f.is_file() and f.suffix == ".py"
                     and os.path.splitext(f.name)[0] not in _AVOID]
            found[d.name] = names
    return found


def available_models() -> dict[str, list[str]]:
    """{family: [model file names]} (methods.py:18-35)."""
    return _scan("models")


def _list_available_models() -> list[str]:
    return [m for names in available_models().values() for m in names]


def available_modules() -> dict[str, list[str]]:
    """{family: [module file names]} (methods.py:58-75)."""
    return _scan("modules")


def load_model(model: str, ver_name: str = None) -> Module:
    """Return the uninstantiated model class (methods.py:78-139)."""
    if ver_name is None:
        ver_name = model
    model = re.sub(r"([a-z])([A-Z])", r"\1_\2", model).lower()
    family = model.split("_")[0].lower()
    source = _PKG / "models" / family / f"{model}.py"
    if not source.exists():
        raise ImportError(f"Model '{model}' not found.")
    try:
        module = importlib.import_module(f"hydrodl2_amd.models.{family}.{model}")
    except ImportError as e:
        raise ImportError(f"Model '{model}' not found.") from e

    cls = getattr(module, ver_name, None)
    if not isinstance(cls, type):
        classes = [a for a in dir(module)
                   if isinstance(getattr(module, a), type)
                   and getattr(module, a).__module__ == module.__name__]
        if not classes:
            raise ImportError(f"Model version '{model}' not found.")
        log.warning(
            f"Model class '{ver_name}' not found in module '{module.__file__}'. "
            f"Falling back to the first available: '{classes[0]}'."
        )
        cls = getattr(module, classes[0])
    return cls


def load_module():
    """methods.py:142-144."""
    raise NotImplementedError("This function is not yet implemented.")
