"""Minimal Hydra/OmegaConf-compatible composer for the task configs.

Hydra and OmegaConf are what the reference uses (isaacgymenvs/__init__.py:8-11,35-37); neither is installed in
this image, and `make()` only ever consumes `cfg.task`, so this module implements exactly the forms the reference
task files use:

  * a root config with a `defaults: [ - task: <Name> ]` list and `key=value` overrides;
  * relative interpolation `${..a}`, `${...a.b}` (n dots = climb n-1 containers) and absolute `${a.b}`;
  * the four custom resolvers registered by the reference: eq, contains, if, resolve_default;
  * nested interpolations inside resolver arguments.

If the real packages are importable they can still be used: `make(cfg=<DictConfig>)` accepts any mapping.
"""
from __future__ import annotations

import copy
import os
import re

import yaml

CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cfg")

_INNER = re.compile(r"\$\{([^${}]*)\}")


def _parse_literal(s):
    if not isinstance(s, str):
        return s
    t = s.strip()
    if len(t) >= 2 and t[0] == t[-1] and t[0] in "\"'":
        return t[1:-1]
    try:
        return yaml.safe_load(t) if t != "" else ""
    except yaml.YAMLError:
        return t


def _split_args(s):
    out, depth, cur, quote = [], 0, "", None
    for ch in s:
        if quote:
            cur += ch
            if ch == quote:
                quote = None
        elif ch in "\"'":
            quote = ch
            cur += ch
        elif ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            depth += ch in "([{"
            depth -= ch in ")]}"
            cur += ch
    out.append(cur)
    return out


RESOLVERS = {
    # reference isaacgymenvs/__init__.py:8-11
    "eq": lambda x, y: str(x).lower() == str(y).lower(),
    "contains": lambda x, y: str(x).lower() in str(y).lower(),
    "if": lambda pred, a, b: a if pred else b,
    "resolve_default": lambda default, arg: default if arg == "" else arg,
}


class _Resolver:
    def __init__(self, root):
        self.root = root

    def lookup(self, path_keys):
        node = self.root
        for k in path_keys:
            if isinstance(node, list):
                node = node[int(k)]
            else:
                node = node[k]
        return node

    def eval_expr(self, expr, container_path):
        expr = expr.strip()
        m = re.match(r"^([A-Za-z_][A-Za-z0-9_]*):(.*)$", expr, re.S)
        if m and m.group(1) in RESOLVERS:
            args = [_parse_literal(a) for a in _split_args(m.group(2))]
            return RESOLVERS[m.group(1)](*args)
        if expr.startswith("."):
            ndots = len(expr) - len(expr.lstrip("."))
            base = list(container_path)
            for _ in range(ndots - 1):
                base.pop()
            keys = base + [k for k in expr.lstrip(".").split(".") if k]
        else:
            keys = expr.split(".")
        # the referenced node may itself be an interpolation
        return self.resolve_value(self.lookup(keys), keys[:-1])

    def resolve_value(self, val, container_path):
        if not isinstance(val, str) or "${" not in val:
            return val
        s = val
        for _ in range(64):
            m = _INNER.search(s)
            if not m:
                break
            r = self.eval_expr(m.group(1), container_path)
            if m.start() == 0 and m.end() == len(s):
                return r
            # embed: strings inside resolver args keep quotes so that ',' splitting stays correct
            rep = r if not isinstance(r, str) else ('"%s"' % r)
            s = s[:m.start()] + str(rep) + s[m.end():]
        return _parse_literal(s) if "${" not in s else s

    def resolve_tree(self, node, path):
        if isinstance(node, dict):
            return {k: self.resolve_tree(v, path + [k]) for k, v in node.items()}
        if isinstance(node, list):
            return [self.resolve_tree(v, path + [str(i)]) for i, v in enumerate(node)]
        return self.resolve_value(node, path[:-1])


def _set_path(d, dotted, value):
    keys = dotted.split(".")
    for k in keys[:-1]:
        d = d.setdefault(k, {})
    d[keys[-1]] = value


def _load_task_yaml(name, cfg_dir):
    path = os.path.join(cfg_dir, "task", name + ".yaml")
    if not os.path.exists(path):
        raise KeyError(f"unknown task config '{name}' (looked in {os.path.dirname(path)})")
    with open(path) as f:
        t = yaml.safe_load(f)
    # task files may inherit: `defaults: [Ant, _self_]` (reference cfg/task/AntSAC.yaml:2-4)
    dl = t.pop("defaults", None)
    if dl:
        # merged in list order, later entries winning; `_self_` is this file's own content (appended when the list does not place it)
        own, out = t, {}
        for item in (list(dl) if "_self_" in dl else list(dl) + ["_self_"]):
            if item == "_self_":
                out = _merge(out, own)
            elif isinstance(item, dict):
                # a config GROUP of the task (reference cfg/task/AllegroKuka.yaml:1-3 `- env: reorientation` -> cfg/task/env/reorientation.yaml):
                # the option's content lands under the group's key
                for group, option in item.items():
                    gpath = os.path.join(cfg_dir, "task", str(group), str(option) + ".yaml")
                    if not os.path.exists(gpath):
                        raise KeyError(f"unknown option '{option}' of task config group '{group}' (looked for {gpath})")
                    with open(gpath) as f:
                        out = _merge(out, {str(group): yaml.safe_load(f) or {}})
            else:
                out = _merge(out, _load_task_yaml(item, cfg_dir))
        t = out
    return t


def _merge(a, b):
    out = copy.deepcopy(a)
    for k, v in b.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = _merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


def compose(config_name="config", overrides=(), cfg_dir=None, resolve=True):
    """Equivalent of `hydra.compose(config_name, overrides)` for the supported forms; returns a plain dict."""
    cfg_dir = cfg_dir or CFG_DIR
    with open(os.path.join(cfg_dir, config_name + ".yaml")) as f:
        root = yaml.safe_load(f)
    groups = {}
    for item in root.pop("defaults", []) or []:
        if isinstance(item, dict):
            groups.update(item)
    plain = []
    for ov in overrides:
        k, _, v = ov.partition("=")
        k = k.lstrip("+")
        if k in ("task", "train", "pbt"):
            groups[k] = v
        else:
            plain.append((k, _parse_literal(v)))
    root["task"] = _load_task_yaml(groups.get("task", "Ant"), cfg_dir)
    for k, v in plain:
        _set_path(root, k, v)
    if not resolve:
        return root
    r = _Resolver(root)
    out = {}
    for k, v in root.items():
        try:
            out[k] = r.resolve_tree(v, [k])
        except (KeyError, TypeError):
            out[k] = v  # learner-side interpolations (e.g. ${train...}) that this engine does not load
    return out


def omegaconf_to_dict(d):
    """reference isaacgymenvs/utils/reformat.py:32-40 -- here configs already are plain resolved dicts."""
    if hasattr(d, "items"):
        return {k: omegaconf_to_dict(v) for k, v in d.items()}
    return d
