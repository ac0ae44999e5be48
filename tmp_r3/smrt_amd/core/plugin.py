"""Plugin discovery with the semantics of smrt/core/plugin.py:13-131: `register_package(pkg)` puts a package in front
of the search order, `import_class(scope, modulename)` returns the first public class defined in
`<pkg>.<scope>.<modulename>`."""
import importlib
import inspect

from .error import SMRTError

user_plugin_package = []
_BUILTIN = "smrt_amd"


def register_package(pkg):
    try:
        importlib.import_module(pkg)
    except ImportError as e:
        raise SMRTError(f"The package must be in the the sys.path list and must contain a __init__.py file. {e}")
    user_plugin_package.insert(0, pkg)


def import_class(scope, modulename, classname=None):
    if "." in modulename:
        candidates = [modulename]
    else:
        candidates = [f"{pkg}.{scope}.{modulename}" for pkg in user_plugin_package + [_BUILTIN]]
    module, err = None, None
    for name in candidates:
        try:
            module = importlib.import_module(name)
            break
        except ImportError as e:
            err = e
    if module is None:
        raise SMRTError(f"Unable to find the module '{modulename}' in the {scope} package(s): {err}")
    if classname is not None:
        return getattr(module, classname)
    for name, obj in inspect.getmembers(module, inspect.isclass):
        if obj.__module__ == module.__name__ and not name.startswith("_"):
            return obj
    raise SMRTError(f"Unable to find a class in the module '{modulename}'")
