"""A small Sacred-compatible configuration layer (sacred is not installed in the image).

Mirrors the subset of Sacred 0.8 the reference's hot-path modules use:
``Ingredient(name)`` with ``@config`` scopes (local variables of the function become entries,
later scopes see earlier entries by parameter name), ``@config_hook``, ``@capture`` (missing
arguments filled by name from the ingredient's config; explicit arguments win), and an
``Experiment`` that merges ingredients, applies ``with a.b=c`` command-line updates and dispatches
``@command`` functions.  Key names and defaults of the reference's ``net``/``data``/``tr``/``te``
ingredients are reproduced by the modules that declare them.
"""
import ast
import functools
import inspect
import sys
import textwrap


def _run_scope(fn, known, fixed=None):
    """Execute a config scope statement by statement and harvest its local variables.

    Like Sacred's ConfigScope: parameters of ``fn`` are filled from ``known`` (earlier entries),
    and a top-level assignment to a name present in ``fixed`` (a ``with k=v`` update) is skipped so
    that later statements of the same scope see the updated value.
    """
    fixed = fixed or {}
    params = list(inspect.signature(fn).parameters)
    tree = ast.parse(textwrap.dedent(inspect.getsource(fn)))
    fdef = next(n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef)))
    glob = dict(fn.__globals__)
    glob.update(inspect.getclosurevars(fn).nonlocals)
    loc = {p: known[p] for p in params if p in known}
    seeded = set(loc)
    fname = inspect.getsourcefile(fn) or "<config>"
    for stmt in fdef.body:
        if isinstance(stmt, ast.Assign) and all(isinstance(t, ast.Name) for t in stmt.targets):
            names = [t.id for t in stmt.targets]
            if any(n in fixed for n in names):
                for n in names:
                    loc[n] = fixed.get(n, loc.get(n))
                    seeded.discard(n)
                continue
            seeded.difference_update(names)
        exec(compile(ast.Module(body=[stmt], type_ignores=[]), fname, "exec"), glob, loc)
    return {k: v for k, v in loc.items() if not k.startswith("_") and k not in seeded}


def _parse_value(text):
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        low = text.lower()
        if low in ("true", "false"):
            return low == "true"
        if low in ("none", "null"):
            return None
        return text


def _set_path(d, dotted, value):
    keys = dotted.split(".")
    for k in keys[:-1]:
        d = d.setdefault(k, {})
    d[keys[-1]] = value


class Ingredient:
    def __init__(self, name, ingredients=(), **_ignored):
        self.name = name
        self.ingredients = list(ingredients)
        self._scopes = []
        self._hooks = []
        self._updates = {}
        self._cfg = None

    # -- declaration ---------------------------------------------------------------------
    def config(self, fn):
        self._scopes.append(fn)
        self._cfg = None
        return fn

    def config_hook(self, fn):
        self._hooks.append(fn)
        return fn

    def add_config(self, **entries):
        self._updates.update(entries)
        self._cfg = None

    # -- resolution ----------------------------------------------------------------------
    @property
    def cfg(self):
        if self._cfg is None:
            cfg = {}
            for scope in self._scopes:
                flat = {k: v for k, v in self._updates.items() if "." not in k}
                cfg.update(_run_scope(scope, {**cfg, **flat}, flat))
            for k, v in self._updates.items():
                _set_path(cfg, k, v)
            self._cfg = cfg
        return self._cfg

    def capture(self, fn):
        sig = inspect.signature(fn)
        ing = self

        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            bound = sig.bind_partial(*args, **kwargs)
            cfg = ing.cfg
            for name, par in sig.parameters.items():
                if name in bound.arguments or par.kind in (par.VAR_POSITIONAL, par.VAR_KEYWORD):
                    continue
                if name in cfg:
                    kwargs[name] = cfg[name]
            return fn(*args, **kwargs)

        return wrapper


class Experiment(Ingredient):
    """Root ingredient with commands and the ``CMD with k=v`` command line."""

    def __init__(self, name, ingredients=(), **kw):
        super().__init__(name, ingredients, **kw)
        self.commands = {}

    def command(self, fn):
        self.commands[fn.__name__] = fn
        return fn

    def full_config(self):
        cfg = dict(self.cfg)
        for ing in self.ingredients:
            cfg[ing.name] = dict(ing.cfg)
        return cfg

    def apply_updates(self, updates):
        by_name = {ing.name: ing for ing in self.ingredients}
        for key, value in updates.items():
            head, _, rest = key.partition(".")
            if head in by_name and rest:
                by_name[head].add_config(**{rest: value})
            else:
                self.add_config(**{key: value})
        cfg = self.full_config()
        for ing in [self] + self.ingredients:
            for hook in ing._hooks:
                out = hook(cfg, None, None)
                if out:
                    (cfg if ing is self else cfg[ing.name]).update(out)
        return cfg

    def run(self, command, config_updates=None):
        cfg = self.apply_updates(config_updates or {})
        fn = self.commands[command]
        sig = inspect.signature(fn)
        kwargs = {}
        for name in sig.parameters:
            if name == "_config":
                kwargs[name] = cfg
            elif name == "_run":
                kwargs[name] = None
            elif name in cfg:
                kwargs[name] = cfg[name]
        return fn(**kwargs)

    def run_commandline(self, argv=None):
        argv = list(sys.argv if argv is None else argv)[1:]
        flags = [a for a in argv if a.startswith("-")]
        argv = [a for a in argv if not a.startswith("-")]
        command = argv[0] if argv else "help"
        updates = {}
        if "with" in argv:
            for item in argv[argv.index("with") + 1:]:
                k, _, v = item.partition("=")
                updates[k] = _parse_value(v)
        if command == "print_config" or "-p" in flags:
            import pprint
            pprint.pprint(self.apply_updates(updates))
            if command == "print_config":
                return None
        if command == "help":
            print("commands:", ", ".join(self.commands))
            return None
        return self.run(command, updates)


# ---------------------------------------------------------------------------------------------
# global / device ingredients of the reference (config.py:13-63): accepted for command-line compatibility
# ---------------------------------------------------------------------------------------------
global_ingredient = Ingredient("g")
device_ingredient = Ingredient("d")


@global_ingredient.config
def global_config():
    model_dir = "model_dir"         # str, directory of checkpoints / file-storage observers
    fileStorage = False             # bool, Sacred FileStorage observer (not built: no observers here)
    mongodb = True                  # bool, Sacred MongoDB observer (not built)
    mongo_port = 7000               # int


@device_ingredient.config
def device_config():
    enable_gpu = True               # bool; the HIP path has no CPU fallback, False raises at model use
    num_threads = 1                 # int, torch CPU threads
    cudnn = {"enabled": True, "benchmark": True}    # accepted, meaningless on MI355X
