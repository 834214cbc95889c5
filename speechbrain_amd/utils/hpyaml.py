"""A self-contained loader for the HyperPyYAML subset that SpeechBrain inference ``hyperparams.yaml``
files use (the ``hyperpyyaml`` package the reference imports in inference/interfaces.py:35 is a separate
distribution and is not required here).

Supported: ``!new:`` (mapping = kwargs, sequence = positional, empty = no arguments), ``!name:``
(callable, partial when arguments are given), ``!apply:``, ``!ref <key>`` (same object instance for every
reference, ``<key[sub]>`` indexing, string interpolation, simple arithmetic), ``!copy <key>``, ``!tuple``,
implicit ``(a, b)`` tuples, ``!PLACEHOLDER``, anchors/aliases, and ``overrides`` (dict or YAML text, merged
recursively: ``{"decoder": {"beam_size": 5}}`` changes one argument of the ``decoder`` object).
Not supported (raises): ``!include:``, ``!module:``/``!import:`` side-effect tags; YAML merge keys (``<<``) are
not expanded.  ``!name:`` without arguments yields the bare callable (hyperpyyaml wraps it in a partial with no
arguments; calling either is the same).

Module paths that start with ``speechbrain.`` are resolved inside ``speechbrain_amd`` -- that is the whole
drop-in: a YAML written for the reference builds the MI355X modules.
"""
import ast
import builtins
import copy
import functools
import importlib
import operator
import re

import yaml

_REF = re.compile(r"<([^<>]+)>")
_TUPLE = re.compile(r"^\(.*\)$")
_OPS = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv,
        ast.FloorDiv: operator.floordiv, ast.Mod: operator.mod, ast.Pow: operator.pow, ast.USub: operator.neg,
        ast.UAdd: operator.pos}


class Placeholder:
    def __repr__(self):
        return "!PLACEHOLDER"


def resolve_name(path: str):
    """'pkg.mod.attr' -> the attribute; ``speechbrain.*`` is served by ``speechbrain_amd.*``."""
    if path == "speechbrain" or path.startswith("speechbrain."):
        path = "speechbrain_amd" + path[len("speechbrain"):]
    parts = path.split(".")
    if len(parts) == 1 and hasattr(builtins, path):
        return getattr(builtins, path)
    for cut in range(len(parts), 0, -1):
        try:
            obj = importlib.import_module(".".join(parts[:cut]))
        except ImportError:
            continue
        try:
            for attr in parts[cut:]:
                obj = getattr(obj, attr)
        except AttributeError as e:
            raise ImportError(f"hyperparams: cannot resolve '{path}' ({e}); the MI355X path implements the "
                              "EncoderDecoderASR modules only") from e
        return obj
    raise ImportError(f"hyperparams: cannot import '{path}'")


def _arith(expr: str):
    def ev(n):
        if isinstance(n, ast.Expression):
            return ev(n.body)
        if isinstance(n, ast.Constant) and isinstance(n.value, (int, float)):
            return n.value
        if isinstance(n, ast.BinOp) and type(n.op) in _OPS:
            return _OPS[type(n.op)](ev(n.left), ev(n.right))
        if isinstance(n, ast.UnaryOp) and type(n.op) in _OPS:
            return _OPS[type(n.op)](ev(n.operand))
        raise ValueError(expr)
    return ev(ast.parse(expr, mode="eval"))


class _Builder:
    def __init__(self, root, overrides):
        self.root = root  # yaml MappingNode
        self.top = {k.value: v for k, v in root.value}
        self.overrides = overrides or {}
        self.memo = {}      # id(node) -> object
        self.building = set()
        self.scalar_loader = yaml.SafeLoader("")

    # ---- references
    def lookup(self, ref: str):
        m = re.match(r"^([^\[\].]+)((?:\[[^\]]+\]|\.[^\[\].]+)*)$", ref.strip())
        if not m:
            raise ValueError(f"hyperparams: bad reference <{ref}>")
        key, rest = m.group(1), m.group(2)
        if key in self.overrides:
            obj = self.overrides[key]
        elif key in self.top:
            obj = self.build(self.top[key])
        else:
            raise KeyError(f"hyperparams: reference to undefined key <{key}>")
        for part in re.findall(r"\[([^\]]+)\]|\.([^\[\].]+)", rest):
            idx = part[0] or part[1]
            if isinstance(obj, (list, tuple)):
                obj = obj[int(idx)]
            elif isinstance(obj, dict):
                obj = obj[idx] if idx in obj else obj[int(idx)]
            else:
                obj = getattr(obj, idx)
        if isinstance(obj, Placeholder):
            raise ValueError(f"hyperparams: <{key}> is a !PLACEHOLDER; pass it through overrides")
        return obj

    def ref(self, text: str):
        text = text.strip()
        whole = _REF.fullmatch(text)
        if whole:
            return self.lookup(whole.group(1))
        values = {}

        def sub(m):
            v = self.lookup(m.group(1))
            values[m.group(0)] = v
            return str(v)
        out = _REF.sub(sub, text)
        if values and all(isinstance(v, (int, float)) and not isinstance(v, bool) for v in values.values()):
            try:
                return _arith(out)
            except (ValueError, SyntaxError):
                pass
        return out

    # ---- construction
    def build(self, node):
        key = id(node)
        if key in self.memo:
            return self.memo[key]
        if key in self.building:
            raise ValueError("hyperparams: circular !ref")
        self.building.add(key)
        try:
            obj = self._build(node)
        finally:
            self.building.discard(key)
        self.memo[key] = obj
        return obj

    def _plain(self, node):
        if isinstance(node, yaml.MappingNode):
            return {self.build(k): self.build(v) for k, v in node.value}
        if isinstance(node, yaml.SequenceNode):
            return [self.build(v) for v in node.value]
        return node.value

    def _args(self, node):
        if isinstance(node, yaml.MappingNode):
            return [], {k.value: self.build(v) for k, v in node.value}
        if isinstance(node, yaml.SequenceNode):
            return [self.build(v) for v in node.value], {}
        if node.value in ("", None):
            return [], {}
        return [self.scalar(node)], {}

    def scalar(self, node, tag=None):
        plain = yaml.ScalarNode(tag or self.scalar_loader.resolve(yaml.ScalarNode, node.value, (True, False)),
                                node.value)
        val = self.scalar_loader.construct_object(plain)
        if isinstance(val, str) and node.style is None and _TUPLE.match(val.strip()):
            try:
                return ast.literal_eval(val.strip())
            except (ValueError, SyntaxError):
                return val
        return val

    def _build(self, node):
        tag = node.tag or ""
        if tag.startswith("!new:") or tag.startswith("!apply:"):
            fn = resolve_name(tag.split(":", 1)[1])
            a, kw = self._args(node)
            return fn(*a, **kw)
        if tag.startswith("!name:"):
            fn = resolve_name(tag.split(":", 1)[1])
            a, kw = self._args(node)
            return functools.partial(fn, *a, **kw) if (a or kw) else fn
        if tag == "!ref":
            return self.ref(node.value)
        if tag == "!copy":
            return copy.deepcopy(self.ref(node.value))
        if tag == "!tuple":
            return tuple(self._plain(node)) if not isinstance(node, yaml.ScalarNode) else ast.literal_eval(node.value)
        if tag == "!PLACEHOLDER":
            return Placeholder()
        if tag.startswith("!include") or tag.startswith("!module") or tag.startswith("!import"):
            raise NotImplementedError(f"hyperparams: the tag {tag} is not supported by this loader")
        if tag.startswith("!") and not tag.startswith("!!"):
            raise NotImplementedError(f"hyperparams: unknown tag {tag}")
        if isinstance(node, yaml.ScalarNode):
            return self.scalar(node, tag if tag.startswith("tag:yaml.org") and node.style is not None else None)
        return self._plain(node)


def _as_node(value):
    """A python override value as a YAML node (so it is built like text that had been in the file).  A string that
    carries a tag ("!ref <x>", "!new:pkg.Class", "!name:...", "!copy <x>", "!tuple (1, 2)") becomes the TAGGED node it
    would have been in the file, as hyperpyyaml resolves it, not a quoted plain string."""
    if isinstance(value, yaml.Node):
        return value
    if isinstance(value, str) and value.startswith("!") and not value.startswith("!!"):
        return yaml.compose(value, Loader=yaml.SafeLoader)
    if isinstance(value, dict):  # values may themselves carry tags: build the mapping node entry by entry
        return yaml.MappingNode("tag:yaml.org,2002:map",
                                [(yaml.ScalarNode("tag:yaml.org,2002:str", str(k)), _as_node(v)) for k, v in value.items()])
    if isinstance(value, (list, tuple)) and any(isinstance(v, (str, dict, list, tuple)) for v in value):
        return yaml.SequenceNode("tag:yaml.org,2002:seq", [_as_node(v) for v in value])
    return yaml.compose(yaml.safe_dump(value, default_flow_style=True), Loader=yaml.SafeLoader)


def _overrides_from_text(text):
    """YAML override text -> {top-level key: node}: composed, not loaded, so that tags inside it survive."""
    root = yaml.compose(text, Loader=yaml.SafeLoader)
    if root is None:
        return {}
    if not isinstance(root, yaml.MappingNode):
        raise ValueError("hyperparams: overrides must be a mapping")
    return {k.value: v for k, v in root.value}


def _merge_overrides(node, overrides, direct):
    """hyperpyyaml's recursive_update on the composed tree: a dict override of a mapping (a plain dict or the
    argument mapping of a !new:/!name:/!apply: object) changes only the named sub-keys; anything else replaces
    the node.  Values that YAML cannot express (live objects) stay in `direct` and replace whole top-level keys."""
    for name, value in overrides.items():
        slot = next((i for i, (k, _) in enumerate(node.value) if k.value == str(name)), None)
        if slot is None:
            key = yaml.ScalarNode("tag:yaml.org,2002:str", str(name))
            try:
                node.value.append((key, _as_node(value)))
            except yaml.YAMLError:
                if direct is None:
                    raise
                direct[name] = value
            continue
        key, child = node.value[slot]
        if isinstance(value, yaml.MappingNode) and isinstance(child, yaml.MappingNode) and not (value.tag or "").startswith("!"):
            value = {k.value: v for k, v in value.value}  # (override text: merge its sub-keys like a dict override)
        if isinstance(value, dict) and isinstance(child, yaml.MappingNode):
            _merge_overrides(child, value, None)
            continue
        try:
            new = _as_node(value)
        except yaml.YAMLError:
            if direct is None:
                raise ValueError(f"hyperparams: the override of '{name}' is not expressible as YAML") from None
            direct[name] = value
            continue
        node.value[slot] = (key, new)


def load_hyperpyyaml(stream, overrides=None):
    """YAML text or file object -> dict of top-level keys with every object constructed.  ``overrides`` (dict or
    YAML text) are merged recursively into the tree before anything is built, like hyperpyyaml does."""
    text = stream if isinstance(stream, str) else stream.read()
    if isinstance(overrides, str):
        overrides = _overrides_from_text(overrides)
    root = yaml.compose(text, Loader=yaml.SafeLoader)
    if root is None:
        return {}
    if not isinstance(root, yaml.MappingNode):
        raise ValueError("hyperparams: the top level must be a mapping")
    direct = {}
    _merge_overrides(root, dict(overrides or {}), direct)
    b = _Builder(root, direct)
    out = {}
    for k, v in root.value:
        name = k.value
        out[name] = b.overrides[name] if name in b.overrides else b.build(v)
    for name, v in out.items():
        if isinstance(v, Placeholder):
            raise ValueError(f"hyperparams: '{name}' is a !PLACEHOLDER; pass it through overrides")
    return out
