"""Configuration + logger bootstrap with the reference's names.

Mirror of pyQSM/set_config.py:16-43: ``config`` is the TOML dict, ``log`` the
``calc`` logger, the file is chosen by ``PY_QSM_CONFIG`` (default: the TOML next
to this module) and the logging set-up by ``PY_QSM_LOG_CONFIG`` (set_config.py:17,
:36-42: a YAML or TOML ``logging.config.dictConfig`` dictionary). Without that
variable the bootstrap pyQSM would have done is kept: the ``log.yml`` beside the
next ``set_config.py`` on ``sys.path`` (pyQSM's own, when this module shadows it),
else the small ``log.yml`` next to this module. Keys consumed by the hot path are
listed in SURVEY.md §8b. Unlike the reference, importing this module prints
nothing and a missing key falls back to the packaged default instead of raising
at import time.
"""
from __future__ import annotations

import logging
import logging.config
import os
import sys

try:  # Python >= 3.11
    import tomllib as _toml
except ModuleNotFoundError:  # pragma: no cover - depends on interpreter
    import tomli as _toml

package_location = os.path.dirname(os.path.abspath(__file__))
_DEFAULT = os.path.join(package_location, "pyqsm_config.toml")
config_file = os.environ.get("PY_QSM_CONFIG", _DEFAULT)


def _default_log_config() -> str:
    """pyQSM's own log.yml when this module shadows pyQSM's set_config.py (the next one on
    sys.path), else the one shipped here."""
    for entry in sys.path:
        base = os.path.abspath(entry or os.getcwd())
        if base != package_location and os.path.isfile(os.path.join(base, "set_config.py")) \
                and os.path.isfile(os.path.join(base, "log.yml")):
            return os.path.join(base, "log.yml")
    return os.path.join(package_location, "log.yml")


log_config_file = os.environ.get("PY_QSM_LOG_CONFIG") or _default_log_config()

log = logging.getLogger()      # until the logging set-up below has run (set_config.py:19)


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


def configure_logging(path: str) -> bool:
    """``logging.config.dictConfig`` from `path` (set_config.py:36-42); on any failure an error
    line and Python's default logging, as in the reference. Returns whether it was applied."""
    cfg = load_config(path)
    try:
        logging.config.dictConfig(cfg)
        return True
    except Exception as error:
        log.error(f"Error loading log config {path}: {error}")
        log.error("Default values will be used")
        return False


log_config_applied = configure_logging(log_config_file)
log = logging.getLogger("calc")

_defaults = load_config(_DEFAULT)
config = _merged(load_config(config_file), _defaults) if config_file != _DEFAULT else _defaults
