"""Configuration + logger bootstrap with the reference's names.

Mirror of pyQSM/set_config.py:16-43: ``config`` is the TOML dict, ``log`` the
``calc`` logger, and the file is chosen by ``PY_QSM_CONFIG`` (default: the TOML
next to this module). Keys consumed by the hot path are listed in SURVEY.md §8b.
Unlike the reference, importing this module prints nothing and a missing key
falls back to the packaged default instead of raising at import time.
"""
from __future__ import annotations

import logging
import os

try:  # Python >= 3.11
    import tomllib as _toml
except ModuleNotFoundError:  # pragma: no cover - depends on interpreter
    import tomli as _toml

package_location = os.path.dirname(os.path.abspath(__file__))
_DEFAULT = os.path.join(package_location, "pyqsm_config.toml")
config_file = os.environ.get("PY_QSM_CONFIG", _DEFAULT)

log = logging.getLogger("calc")


def load_config(path: str) -> dict:
    """TOML (or YAML, by extension) -> dict; {} with an error log on failure,
    like pyQSM/set_config.py:21-33."""
    try:
        if path.endswith((".yml", ".yaml")):
            import yaml
            with open(path) as f:
                return yaml.safe_load(f) or {}
        with open(path, "rb") as f:
            return _toml.load(f)
    except Exception as error:  # same behaviour as the reference: log and go on
        log.error(f"Error loading config {path}: {error}")
        log.error("Default values will be used")
        return {}


def _merged(user: dict, default: dict) -> dict:
    out = {k: (dict(v) if isinstance(v, dict) else v) for k, v in default.items()}
    for k, v in user.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k].update(v)
        else:
            out[k] = v
    return out


_defaults = load_config(_DEFAULT)
config = _merged(load_config(config_file), _defaults) if config_file != _DEFAULT else _defaults
