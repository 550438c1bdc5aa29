"""Import alias: the package directory is ``yolo-nano_amd/`` (not a valid
Python identifier), so ``import yolo_nano_amd`` resolves its submodules there."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "yolo-nano_amd")]
_init = _os.path.join(__path__[0], "__init__.py")
exec(compile(open(_init).read(), _init, "exec"))
