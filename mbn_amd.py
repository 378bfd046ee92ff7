"""Import shim: the package directory name (`cnn-mobilenet-v1-implementation-on-aws-fpga-using-opencl_amd`)
is not a valid Python identifier, so load it by path under the module name `mbn_amd_pkg`."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "cnn-mobilenet-v1-implementation-on-aws-fpga-using-opencl_amd")


def import_package():
    name = "mbn_amd_pkg"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(_PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod
