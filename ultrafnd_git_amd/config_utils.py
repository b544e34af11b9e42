"""YAML config loading with the reference's semantics (src/utils/config_utils.py:23-86):
missing file / unreadable YAML / non-dict content -> {} (callers fall back to their in-code
defaults), paths tried relative to the cwd and then to the repository root, results cached."""
from __future__ import annotations

from pathlib import Path
from typing import Any, Dict, Optional

import yaml

_REPO_ROOT = Path(__file__).resolve().parents[1]


class ConfigManager:
    def __init__(self):
        self._cache: Dict[str, Dict[str, Any]] = {}

    def load_config(self, path: str, defaults: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
        p = Path(path)
        if not p.exists() and (_REPO_ROOT / path).exists():
            p = _REPO_ROOT / path
        key = str(p.resolve()) if p.exists() else str(Path(path))
        if key not in self._cache:
            cfg: Dict[str, Any] = {}
            if p.is_file():
                try:
                    loaded = yaml.safe_load(p.read_text(encoding="utf-8"))
                    cfg = loaded if isinstance(loaded, dict) else {}
                except Exception:
                    cfg = {}
            self._cache[key] = cfg
        cfg = self._cache[key]
        if not defaults:
            return cfg
        out = dict(defaults)
        out.update(cfg)
        return out


def load_yaml(path: str, defaults: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
    return ConfigManager().load_config(path, defaults=defaults)
