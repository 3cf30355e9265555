"""Import shim: the package sources live in ``pulser-diff_amd/`` (a directory name Python cannot import
directly); this module makes them importable as ``pulser_diff_amd``."""
import pathlib as _pathlib

_real = _pathlib.Path(__file__).resolve().parent.parent / "pulser-diff_amd"
__path__ = [str(_real)]
exec(compile((_real / "__init__.py").read_text(), str(_real / "__init__.py"), "exec"))
