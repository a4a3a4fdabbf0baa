"""Importable alias for the package directory ``multimodal-gesture-recognition-with-lstms-and-ctc_amd/``
(its mandated name is not a valid Python identifier).  ``import mgr_amd`` loads that directory as the
package ``mgr_amd``; sub-modules resolve normally (``mgr_amd.multimodal_fusion.multimodal`` ...)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multimodal-gesture-recognition-with-lstms-and-ctc_amd")
_spec = importlib.util.spec_from_file_location("mgr_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mgr_amd"] = _mod
_spec.loader.exec_module(_mod)
