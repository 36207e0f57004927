"""Load the REAL reference (mct10/Speech-Vecalign) in this container, for pinning the oracle.

TEST INFRASTRUCTURE ONLY.  The reference lives read-only under /root/reference and exists
only in the build container: it never travels to the GPU box, and nothing in the product
(`speech-vecalign_amd/`) imports this file.  Users: tests/golden/make_golden.py (fixture
generation) and tests/test_oracle_vs_reference.py (skipped when the reference is absent).

How: the reference compiles its Cython module at import through pyximport with inplace=True
(/root/reference/svecalign/vecalign/dp_utils.py:23-27), which cannot write under the read-only
tree.  oracle/build_ref.sh performs that same Cython->gcc step into oracle/_ref/; here the
resulting module is registered as `svecalign.vecalign.dp_core` *before* dp_utils is imported,
so pyximport's hook never fires and no reference source is copied anywhere.
"""
import importlib.machinery
import importlib.util
import os
import subprocess
import sys
import sysconfig
import types

REF_ROOT = os.environ.get("SVX_REFERENCE", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "svecalign", "vecalign", "dp_core.pyx"))


def load():
    """Return the reference's modules as a namespace: .dp_core .dp_utils .vecalign .embedding_utils ..."""
    if not available():
        raise RuntimeError("reference not present at %s" % REF_ROOT)
    if "svecalign.vecalign.dp_utils" not in sys.modules:
        subprocess.check_call(["bash", os.path.join(_HERE, "build_ref.sh")])
        so = os.path.join(_HERE, "_ref", "dp_core" + sysconfig.get_config_var("EXT_SUFFIX"))
        sys.dont_write_bytecode = True
        if REF_ROOT not in sys.path:
            sys.path.insert(0, REF_ROOT)
        import svecalign.vecalign  # noqa: F401  (package __init__ is empty)
        loader = importlib.machinery.ExtensionFileLoader("svecalign.vecalign.dp_core", so)
        spec = importlib.util.spec_from_file_location("svecalign.vecalign.dp_core", so, loader=loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
        sys.modules["svecalign.vecalign.dp_core"] = mod
        # audio_utils imports soundfile at module top (only a constant is used on this path)
        if "soundfile" not in sys.modules:
            try:
                import soundfile  # noqa: F401
            except Exception:
                sys.modules["soundfile"] = types.ModuleType("soundfile")
    ns = types.SimpleNamespace()
    import svecalign.vecalign.dp_core as dp_core
    import svecalign.vecalign.dp_utils as dp_utils
    import svecalign.vecalign.vecalign as vecalign
    import svecalign.vecalign.score as score
    import svecalign.utils.embedding_utils as embedding_utils
    import svecalign.utils.file_utils as file_utils
    ns.dp_core, ns.dp_utils, ns.vecalign, ns.score = dp_core, dp_utils, vecalign, score
    ns.embedding_utils, ns.file_utils = embedding_utils, file_utils
    ns.root = REF_ROOT
    return ns
