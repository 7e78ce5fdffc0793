"""``tf.app.flags``-style flag registry.

Reproduces the surface used at /root/reference/example.py:56,71-105: ``DEFINE_string`` /
``DEFINE_integer`` (plus float/boolean), a module-level ``FLAGS`` object that parses
``sys.argv`` lazily on first attribute access, and typed conversion of *defaults* (the
reference passes the raw ``TASK_INDEX`` environment string as the integer flag's default
and then relies on ``FLAGS.task_index == 0``, /root/reference/example.py:61,73,190).

Precedence: value on the command line > default given to ``DEFINE_*`` (which the scripts
fill from the environment).  Unknown command-line arguments are left alone.
"""
from __future__ import annotations

import sys
from typing import Any, Callable, Dict, List, Optional


def _to_bool(v) -> bool:
    if isinstance(v, bool):
        return v
    if isinstance(v, str):
        s = v.strip().lower()
        if s in ("1", "true", "t", "yes", "y"):
            return True
        if s in ("0", "false", "f", "no", "n", ""):
            return False
        raise ValueError("not a boolean: %r" % v)
    return bool(v)


class _Flag:
    __slots__ = ("name", "default", "help", "parser", "kind")

    def __init__(self, name: str, default: Any, help: str, parser: Callable, kind: str):
        self.name, self.help, self.parser, self.kind = name, help, parser, kind
        self.default = None if default is None else parser(default)


class FlagValues:
    """Attribute-style access to flag values; parses the command line on first use."""

    def __init__(self):
        self.__dict__["_flags"] = {}
        self.__dict__["_values"] = {}
        self.__dict__["_parsed"] = False
        self.__dict__["_unparsed"] = []

    # -- registration
    def _define(self, name: str, default, help: str, parser: Callable, kind: str):
        flags: Dict[str, _Flag] = self.__dict__["_flags"]
        if name in flags:
            raise ValueError("flag --%s defined twice" % name)
        flags[name] = _Flag(name, default, help, parser, kind)
        self.__dict__["_parsed"] = False

    # -- parsing
    def _parse(self, argv: Optional[List[str]] = None):
        flags: Dict[str, _Flag] = self.__dict__["_flags"]
        values = {n: f.default for n, f in flags.items()}
        args = list(sys.argv[1:] if argv is None else argv)
        rest: List[str] = []
        i = 0
        while i < len(args):
            a = args[i]
            if a == "--":
                rest.extend(args[i + 1:])
                break
            if a.startswith("--"):
                body = a[2:]
                if "=" in body:
                    name, val = body.split("=", 1)
                    has_val = True
                else:
                    name, val, has_val = body, None, False
                name = name.replace("-", "_")
                if name in flags:
                    f = flags[name]
                    if f.kind == "bool" and not has_val:
                        values[name] = True
                    else:
                        if not has_val:
                            i += 1
                            if i >= len(args):
                                raise ValueError("flag --%s needs a value" % name)
                            val = args[i]
                        values[name] = f.parser(val)
                    i += 1
                    continue
                if name.startswith("no") and name[2:] in flags and flags[name[2:]].kind == "bool" and not has_val:
                    values[name[2:]] = False
                    i += 1
                    continue
            rest.append(a)
            i += 1
        self.__dict__["_values"] = values
        self.__dict__["_unparsed"] = rest
        self.__dict__["_parsed"] = True
        return rest

    def __call__(self, argv: Optional[List[str]] = None):
        return self._parse(argv)

    def __getattr__(self, name: str):
        if name.startswith("_"):
            raise AttributeError(name)
        if not self.__dict__["_parsed"]:
            self._parse()
        values = self.__dict__["_values"]
        if name not in values:
            raise AttributeError("unknown flag %r" % name)
        return values[name]

    def __setattr__(self, name: str, value):
        if not self.__dict__["_parsed"]:
            self._parse()
        flags = self.__dict__["_flags"]
        if name not in flags:
            raise AttributeError("unknown flag %r" % name)
        self.__dict__["_values"][name] = None if value is None else flags[name].parser(value)

    def flag_values_dict(self) -> Dict[str, Any]:
        if not self.__dict__["_parsed"]:
            self._parse()
        return dict(self.__dict__["_values"])

    def help_text(self) -> str:
        lines = []
        for f in self.__dict__["_flags"].values():
            lines.append("  --%s (%s, default %r)\n      %s" % (f.name, f.kind, f.default, f.help))
        return "\n".join(lines)

    def _reset(self):
        """Forget every definition (used by tests)."""
        self.__dict__["_flags"] = {}
        self.__dict__["_values"] = {}
        self.__dict__["_parsed"] = False


FLAGS = FlagValues()


def DEFINE_string(name: str, default: Optional[str], help: str = "", flag_values: FlagValues = FLAGS):
    flag_values._define(name, default, help, str, "string")


def DEFINE_integer(name: str, default, help: str = "", flag_values: FlagValues = FLAGS):
    flag_values._define(name, default, help, int, "int")


def DEFINE_float(name: str, default, help: str = "", flag_values: FlagValues = FLAGS):
    flag_values._define(name, default, help, float, "float")


def DEFINE_boolean(name: str, default, help: str = "", flag_values: FlagValues = FLAGS):
    flag_values._define(name, default, help, _to_bool, "bool")


DEFINE_bool = DEFINE_boolean
