"""Python config files the way the reference's entry scripts read them.

``infer_wedetect.py:150-153`` does ``cfg = Config.fromfile(args.config)`` then
``cfg.merge_from_dict(args.cfg_options)`` with ``--cfg-options`` parsed by ``DictAction``
(infer_wedetect.py:88-97); the files are ``config/wedetect_{tiny,base,large}.py`` with
``_base_ = ["default_runtime.py"]``.  mmengine is not a dependency of this package, so the three
pieces those lines need are built here with the same observable behaviour:

  * ``Config.fromfile(path)``: the file is plain Python; ``_base_`` (a path or a list of paths, relative
    to the file) is loaded first and the file's own top-level names are merged over it — dicts
    recursively, ``_delete_=True`` in a child dict replaces instead of merging, everything else
    overrides.  ``_base_.name`` inside the file resolves to the merged base value.  Modules,
    functions and classes defined at top level are not part of the config.  ``custom_imports`` is
    honoured (``import_custom_modules=True``): with ``allow_failed_imports=False`` a missing module
    raises ``ImportError``.
  * attribute and item access on nested dicts (``cfg.model.test_cfg.score_thr``), ``cfg.get``,
    assignment (``cfg.work_dir = ...``), ``cfg.to_dict()``.
  * ``merge_from_dict({'model.test_cfg.score_thr': 0.05})``: dotted keys, integer components index
    into lists.

Not implemented (no shipped config uses them): ``{{ fileDirname }}`` template variables, lazy-import
configs, ``.json`` / ``.yaml`` files, ``_base_`` across packages (``mmdet::...``).  Using one raises.
"""
from __future__ import annotations

import argparse
import ast
import copy
import importlib
import os
import types
from typing import Any, Dict, Iterable, List, Optional, Sequence, Union

BASE_KEY = "_base_"
DELETE_KEY = "_delete_"
RESERVED = ("filename", "text", "pretty_text", "env_variables")


class ConfigDict(dict):
    """dict with attribute access; nested dicts (also inside lists / tuples) are converted on the way in.
    A missing attribute raises ``AttributeError`` (what ``hasattr`` / ``getattr(default)`` need)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, ConfigDict):
            return v
        if isinstance(v, dict):
            return ConfigDict(v)
        if isinstance(v, list):
            return [ConfigDict._wrap(x) for x in v]
        if isinstance(v, tuple):
            return tuple(ConfigDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(f"'ConfigDict' object has no attribute {k!r}") from None

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        try:
            del self[k]
        except KeyError:
            raise AttributeError(k) from None

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def copy(self):
        return ConfigDict(self)

    def to_dict(self) -> dict:
        return _plain(self)


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_plain(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_plain(x) for x in v)
    return v


def _merge(base: dict, child: dict, allow_list_keys: bool = False) -> dict:
    """``child`` over ``base``: dict values merge recursively unless the child carries ``_delete_=True``; with
    ``allow_list_keys`` a child dict whose keys are all digit strings patches the elements of a base list."""
    out = copy.deepcopy(base)
    if allow_list_keys and isinstance(out, list):
        for k, v in child.items():
            if not k.isdigit() or int(k) >= len(out):
                raise KeyError(f"index {k} is out of range for a list of {len(out)} entries")
            i = int(k)
            out[i] = _merge(out[i], v, True) if isinstance(v, dict) and isinstance(out[i], (dict, list)) else v
        return out
    for k, v in child.items():
        if isinstance(v, dict):
            v = dict(v)
            delete = v.pop(DELETE_KEY, False)
            if k in out and not delete and isinstance(out[k], dict):
                out[k] = _merge(out[k], v, allow_list_keys)
                continue
            if k in out and not delete and allow_list_keys and isinstance(out[k], list):
                out[k] = _merge(out[k], v, True)
                continue
            if k in out and not delete and not isinstance(out[k], dict):
                raise TypeError(f"{k}={v} in the child config cannot inherit from the base because {k} is a dict in "
                                f"the child but {type(out[k]).__name__} in the base; set {DELETE_KEY}=True to replace it")
            out[k] = _merge({}, v, allow_list_keys)
        else:
            out[k] = copy.deepcopy(v)
    return out


class _BaseRef(dict):
    """What ``_base_`` names inside a config file while it executes: attribute access into the merged bases."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(f"_base_ has no variable {k!r}") from None
        return _BaseRef(v) if isinstance(v, dict) and not isinstance(v, _BaseRef) else v


def _file2dict(filename: str, seen: Sequence[str] = ()) -> Dict[str, Any]:
    filename = os.path.abspath(os.path.expanduser(filename))
    if not os.path.isfile(filename):
        raise FileNotFoundError(f"config file {filename!r} does not exist")
    if not filename.endswith(".py"):
        raise OSError("only .py config files are supported")
    if filename in seen:
        raise RecursionError(f"circular _base_ chain through {filename}")
    with open(filename, encoding="utf-8") as f:
        text = f.read()
    if "{{" in text and "}}" in text:
        raise NotImplementedError(f"{filename}: '{{{{ ... }}}}' template variables are not supported")
    tree = ast.parse(text, filename=filename)
    bases: List[str] = []
    body = []
    for node in tree.body:
        if (isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name)
                and node.targets[0].id == BASE_KEY):
            val = ast.literal_eval(node.value)
            bases = [val] if isinstance(val, str) else list(val)
            continue
        body.append(node)
    tree.body = body
    merged: Dict[str, Any] = {}
    for b in bases:
        if "::" in b:
            raise NotImplementedError(f"{filename}: cross-package _base_ ({b}) is not supported")
        sub = _file2dict(os.path.join(os.path.dirname(filename), b), tuple(seen) + (filename,))
        dup = set(merged) & set(sub)
        if dup:
            raise KeyError(f"duplicate key(s) {sorted(dup)} in the _base_ files of {filename}")
        merged.update(sub)
    scope: Dict[str, Any] = {"__file__": filename, "__name__": "__wedetect_config__", BASE_KEY: _BaseRef(merged)}
    exec(compile(tree, filename, "exec"), scope)          # config files are code, exactly as for mmengine
    own = {k: _plain(v) for k, v in scope.items()
           if not k.startswith("__") and k != BASE_KEY
           and not isinstance(v, (types.ModuleType, types.FunctionType, type))}
    for k in own:
        if k in RESERVED:
            raise KeyError(f"{k} is reserved for the config object")
    return _merge(merged, own)


class Config:
    """``Config.fromfile(path)`` / ``Config(dict)``; see the module docstring."""

    def __init__(self, cfg_dict: Optional[dict] = None, filename: Optional[str] = None):
        if cfg_dict is None:
            cfg_dict = {}
        if not isinstance(cfg_dict, dict):
            raise TypeError(f"cfg_dict must be a dict, got {type(cfg_dict).__name__}")
        object.__setattr__(self, "_cfg_dict", ConfigDict(cfg_dict))
        object.__setattr__(self, "_filename", filename)

    @staticmethod
    def fromfile(filename: Union[str, os.PathLike], import_custom_modules: bool = True) -> "Config":
        filename = str(filename)
        d = _file2dict(filename)
        if import_custom_modules and d.get("custom_imports"):
            ci = d["custom_imports"]
            imports = ci.get("imports", [])
            for name in ([imports] if isinstance(imports, str) else imports):
                try:
                    importlib.import_module(name)
                except ImportError:
                    if not ci.get("allow_failed_imports", False):
                        raise
        return Config(d, filename=filename)

    @property
    def filename(self):
        return self._filename

    def merge_from_dict(self, options: Dict[str, Any], allow_list_keys: bool = True) -> None:
        nested: Dict[str, Any] = {}
        for full, v in options.items():
            d = nested
            parts = full.split(".")
            for p in parts[:-1]:
                d = d.setdefault(p, {})
            d[parts[-1]] = v
        object.__setattr__(self, "_cfg_dict", ConfigDict(_merge(self._cfg_dict.to_dict(), nested, allow_list_keys)))

    def to_dict(self) -> dict:
        return self._cfg_dict.to_dict()

    def get(self, k, default=None):
        return self._cfg_dict.get(k, default)

    def keys(self):
        return self._cfg_dict.keys()

    def items(self):
        return self._cfg_dict.items()

    def __contains__(self, k):
        return k in self._cfg_dict

    def __len__(self):
        return len(self._cfg_dict)

    def __iter__(self):
        return iter(self._cfg_dict)

    def __getattr__(self, k):
        return getattr(self._cfg_dict, k)

    def __getitem__(self, k):
        return self._cfg_dict[k]

    def __setattr__(self, k, v):
        self._cfg_dict[k] = v

    __setitem__ = __setattr__

    def __repr__(self):
        return f"Config (path: {self._filename}): {self._cfg_dict!r}"

    def __deepcopy__(self, memo):
        return Config(copy.deepcopy(self._cfg_dict.to_dict(), memo), self._filename)


class DictAction(argparse.Action):
    """``--cfg-options key=value [key=value ...]``: values parse as int, float, bool, None or str; ``a,b`` and
    ``[a,b]`` become lists, ``(a,b)`` tuples, nested brackets allowed (``key="[(a,b),(c,d)]"``)."""

    @staticmethod
    def _scalar(s: str):
        for cast in (int, float):
            try:
                return cast(s)
            except ValueError:
                pass
        low = s.lower()
        if low in ("true", "false"):
            return low == "true"
        if s == "None":
            return None
        return s

    @staticmethod
    def _split_top(s: str) -> List[str]:
        """Splits at commas outside any bracket pair."""
        if s.count("(") != s.count(")") or s.count("[") != s.count("]"):
            raise ValueError(f"imbalanced brackets in {s!r}")
        parts, depth, cur = [], 0, []
        for ch in s:
            if ch in "([":
                depth += 1
            elif ch in ")]":
                depth -= 1
            if ch == "," and depth == 0:
                parts.append("".join(cur))
                cur = []
            else:
                cur.append(ch)
        parts.append("".join(cur))
        return parts

    @classmethod
    def _value(cls, s: str):
        s = s.strip("'\"").replace(" ", "")
        is_tuple = False
        if s.startswith("(") and s.endswith(")"):
            is_tuple, s = True, s[1:-1]
        elif s.startswith("[") and s.endswith("]"):
            s = s[1:-1]
        elif "," not in s:
            return cls._scalar(s)
        vals = [cls._value(p) for p in cls._split_top(s) if p != ""]
        return tuple(vals) if is_tuple else vals

    def __call__(self, parser, namespace, values, option_string=None):
        options = copy.copy(getattr(namespace, self.dest, None) or {})
        for kv in values or []:
            if "=" not in kv:
                raise argparse.ArgumentError(self, f"expected key=value, got {kv!r}")
            k, v = kv.split("=", maxsplit=1)
            options[k] = self._value(v)
        setattr(namespace, self.dest, options)
