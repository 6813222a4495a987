"""Registries: config ``type`` names -> the objects of this package.

The reference builds everything through mmengine registries (``MODELS.build(cfg.model)`` inside
``init_detector``, ``TRANSFORMS`` inside ``Compose(cfg.test_pipeline)``; its classes register with
``@MODELS.register_module()``, e.g. yolo_world.py:11, mm_backbone.py:330, yolo_world_pafpn.py:987).
``MODELS`` / ``TRANSFORMS`` here have the same build contract — ``build(cfg)`` pops ``type`` (and the
``_scope_`` hint), looks the name up, calls it with the remaining keys; an unknown name raises
``KeyError`` — and ``register_with_mmengine()`` additionally enters the same names into mmdet's /
mmengine's own registries when those packages are importable, so a stock ``init_detector`` resolves them
to this package too (``custom_imports = dict(imports=["wedetect"])``, config/wedetect_base.py:37, imports the
``wedetect`` shim at the repository root, which calls it).
"""
from __future__ import annotations

import inspect
from typing import Any, Callable, Dict, Optional


class Registry:
    def __init__(self, name: str):
        self.name = name
        self._modules: Dict[str, Callable] = {}

    def __contains__(self, key: str) -> bool:
        return key in self._modules

    def __len__(self) -> int:
        return len(self._modules)

    @property
    def module_dict(self) -> Dict[str, Callable]:
        return dict(self._modules)

    def get(self, key: str) -> Optional[Callable]:
        if "." in key:                     # "mmdet.LoadAnnotations": scope prefix
            key = key.split(".")[-1]
        return self._modules.get(key)

    def register_module(self, name: Optional[str] = None, force: bool = False, module: Optional[Callable] = None):
        """Decorator (``@REG.register_module()``) or direct call (``REG.register_module(module=cls)``)."""
        def _add(obj):
            key = name or obj.__name__
            if key in self._modules and not force and self._modules[key] is not obj:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._modules[key] = obj
            return obj
        if module is not None:
            return _add(module)
        return _add

    def build(self, cfg: dict, **default_args) -> Any:
        if not isinstance(cfg, dict):
            raise TypeError(f"cfg should be a dict, got {type(cfg).__name__}")
        if "type" not in cfg:
            raise KeyError(f"`cfg` must contain the key 'type', got {sorted(cfg)}")
        args = dict(cfg)
        for k, v in default_args.items():
            args.setdefault(k, v)
        t = args.pop("type")
        args.pop("_scope_", None)
        if callable(t) and not isinstance(t, str):
            obj = t
        else:
            obj = self.get(t)
            if obj is None:
                raise KeyError(f"{t} is not in the {self.name} registry of wedetect_amd. The shipped configs' names are "
                               f"registered; training-only and dataset classes are out of this package's scope")
        try:
            return obj(**args)
        except TypeError as e:
            where = f"{inspect.getsourcefile(obj)}" if inspect.isclass(obj) or inspect.isfunction(obj) else repr(obj)
            raise type(e)(f"{t} ({where}): {e}") from e


MODELS = Registry("model")
TRANSFORMS = Registry("transform")


def register_with_mmengine() -> bool:
    """Enters every name of ``MODELS`` / ``TRANSFORMS`` into mmdet's registries (children of mmengine's root ones,
    scope "mmdet" — the ``default_scope`` of config/default_runtime.py) if they can be imported; returns whether
    that happened.  ``force=True``: this package replaces same-named reference classes on purpose."""
    try:
        from mmdet.registry import MODELS as MM_MODELS
        from mmdet.registry import TRANSFORMS as MM_TRANSFORMS
    except Exception:
        try:
            from mmengine.registry import MODELS as MM_MODELS
            from mmengine.registry import TRANSFORMS as MM_TRANSFORMS
        except Exception:
            return False
    for src, dst in ((MODELS, MM_MODELS), (TRANSFORMS, MM_TRANSFORMS)):
        for name, obj in src.module_dict.items():
            dst.register_module(name=name, force=True, module=obj)
    return True
