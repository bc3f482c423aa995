import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def load_package():
    """Imports crust-render_amd/ (hyphenated directory) as the module `crust_render_amd`."""
    import importlib.util
    if "crust_render_amd" in sys.modules:
        return sys.modules["crust_render_amd"]
    path = os.path.join(ROOT, "crust-render_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("crust_render_amd", path,
                                                  submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["crust_render_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def crt():
    mod = load_package()
    mod.lib()  # raises if libcrt_amd.so has not been built: no fallback
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import ora
    ora.build()
    return ora
