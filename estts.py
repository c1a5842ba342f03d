"""Import alias: `import estts` loads the package kept in the directory
`expressive-speech-synthesis-research_amd/` (whose name is not a valid Python
identifier) under the module name `estts`."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'expressive-speech-synthesis-research_amd')
_spec = importlib.util.spec_from_file_location(
    'estts', os.path.join(_pkg_dir, '__init__.py'), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['estts'] = _mod
_spec.loader.exec_module(_mod)
