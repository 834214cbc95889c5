"""speechbrain.utils.parameter_transfer mirror: ``Pretrainer`` for LOCAL sources
(utils/parameter_transfer.py:33-360).  ``collect_files`` resolves ``<source>/<name>.ckpt`` (or the
explicit ``paths`` entry) on the filesystem; HuggingFace / URL fetching is not part of the MI355X
path (no network on the box) and raises with the path it looked for.  ``load_collected`` applies the
reference's transfer hooks (utils/checkpoints.py:236-304): non-strict ``load_state_dict`` for
torch modules, ``_load`` for objects that define it (InputNormalization), ``load`` for
SentencePieceProcessor.
"""
import logging
import os
import pathlib

import torch

logger = logging.getLogger(__name__)


def torch_parameter_transfer(obj, path):
    """Non-strict state_dict load with a warning per missing / unexpected key (checkpoints.py:236-265)."""
    state_dict = torch.load(str(path), map_location="cpu")
    res = obj.load_state_dict(state_dict, strict=False)
    for k in res.missing_keys:
        logger.warning(f"During parameter transfer to {type(obj).__name__} loading from {path}, the transferred "
                       f"parameters did not have parameters for the key: {k}")
    for k in res.unexpected_keys:
        logger.warning(f"During parameter transfer to {type(obj).__name__} loading from {path}, the object could "
                       f"not use the parameters loaded with the key: {k}")


class Pretrainer:
    def __init__(self, collect_in=None, loadables=None, paths=None, custom_hooks=None, conditions=None):
        self.loadables, self.paths, self.custom_hooks, self.conditions = {}, {}, {}, {}
        self.set_collect_in(collect_in)
        self.add_loadables(loadables or {})
        self.add_paths(paths or {})
        self.add_custom_hooks(custom_hooks or {})
        self.add_conditions(conditions or {})
        self.collected = {}

    def set_collect_in(self, path):
        self.collect_in = pathlib.Path(path) if path is not None else None

    def add_loadables(self, loadables):
        self.loadables.update(loadables)

    def add_paths(self, paths):
        self.paths.update(paths)

    def add_custom_hooks(self, custom_hooks):
        self.custom_hooks.update(custom_hooks)

    def add_conditions(self, conditions):
        self.conditions.update(conditions)

    @staticmethod
    def split_path(path):
        """'src/dir/file.ckpt' -> ('src/dir', 'file.ckpt') (parameter_transfer.py:155-186)."""
        path = str(path)
        if "/" not in path:
            return "./", path
        source, filename = path.rsplit("/", maxsplit=1)
        return source, filename

    def is_loadable(self, name):
        if name not in self.conditions:
            return True
        cond = self.conditions[name]
        return bool(cond() if callable(cond) else cond)

    def collect_files(self, default_source=None, **unused_fetch_options):
        """Resolve every loadable to an existing local file; returns {name: path}."""
        self.collected = {}
        for name in self.loadables:
            if not self.is_loadable(name):
                continue
            if name in self.paths:
                source, filename = self.split_path(self.paths[name])
            elif default_source is not None:
                source, filename = str(default_source), name + ".ckpt"
            else:
                raise ValueError(f"Path not specified for '{name}', and no default_source given!")
            path = os.path.join(source, filename)
            if not os.path.isfile(path):
                raise FileNotFoundError(
                    f"Pretrainer: '{path}' does not exist. Only local sources are supported on the MI355X path "
                    "(there is no HuggingFace / URL fetching); download the model directory first.")
            self.collected[name] = pathlib.Path(path)
        return dict(self.collected)

    def load_collected(self):
        logger.info(f"Loading pretrained files for: {', '.join(self.loadables)}")
        for name, obj in self.loadables.items():
            if not self.is_loadable(name):
                continue
            if name not in self.collected:
                raise RuntimeError(f"Pretrainer: call collect_files() before load_collected() ('{name}' is not collected)")
            path = self.collected[name]
            if name in self.custom_hooks:
                self.custom_hooks[name](obj, path)
            elif hasattr(obj, "_load") and callable(obj._load):  # mark_as_transfer methods (InputNormalization)
                obj._load(path)
            elif isinstance(obj, torch.nn.Module):
                torch_parameter_transfer(obj, path)
            elif hasattr(obj, "load") and type(obj).__name__ == "SentencePieceProcessor":
                obj.load(str(path))
            else:
                raise RuntimeError(f"Don't know how to load {type(obj)}. Register a custom hook for '{name}'.")
