"""CPU-only: the C-ABI shared library loads and exports every entry point include/estd_hip.h declares
(no compute calls without a GPU), and the ctypes mirror of the descriptor struct has the C layout."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from estdepth_amd import build
    return build.build()


def _declared(with_ab=False):
    """entry points the header declares; the #ifdef ESTD_BUILD_AB blocks (superseded A/B kernels) only for a library built that way"""
    src = open(os.path.join(ROOT, "include", "estd_hip.h")).read()
    if not with_ab:
        src = re.sub(r"#ifdef ESTD_BUILD_AB.*?#endif\n", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(estd_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(libpath):
    from estdepth_amd import _native
    handle = ctypes.CDLL(libpath)
    names = _declared(with_ab=_native.has_ab())
    assert len(names) >= 18
    for n in names:
        assert hasattr(handle, n), "missing export: " + n


def test_python_binding_covers_header():
    from estdepth_amd import _native
    assert sorted(_native.EXPORTED_SYMBOLS) == _declared()
    assert sorted(_native.EXPORTED_SYMBOLS + _native.AB_SYMBOLS) == _declared(with_ab=True)


def test_version_and_status_strings(libpath):
    from estdepth_amd import _native
    lib = _native.lib()
    assert lib.estd_version() >= 100
    assert lib.estd_status_string(0) == b"ok"
    assert b"argument" in lib.estd_status_string(-1)


def test_argument_validation_without_gpu(libpath):
    """Entry points validate before launching: null pointers / bad sizes return ESTD_ERR_ARG (no GPU needed)."""
    from estdepth_amd import _native
    lib = _native.lib()
    assert lib.estd_conv3d_k3(None, None) == -1
    assert lib.estd_conv3d_k3_grid(1, 64, 120, 160) == 64 * 15 * 10
    assert lib.estd_conv3d_k3_grid(0, 1, 1, 1) == -1
    assert lib.estd_softargmin_up(None, None, None, None, 1, 1, 1, 1, 4, None) == -1
    assert lib.estd_homo_warp_costvol(None, None, None, None, None, 1, 1, 1, None) == -1
    d = _native.Conv3dDesc()
    assert lib.estd_conv3d_k3(ctypes.byref(d), None) == -1


def test_desc_struct_layout(libpath, tmp_path):
    """sizeof/offsetof of estd_conv3d_desc as the C compiler sees it == the ctypes mirror."""
    from estdepth_amd import _native
    src = tmp_path / "layout.c"
    fields = [f[0] for f in _native.Conv3dDesc._fields_]
    body = "\n".join('printf("%%zu\\n", offsetof(estd_conv3d_desc, %s));' % f for f in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "estd_hip.h"\nint main(){printf("%zu\\n", sizeof(estd_conv3d_desc));\n'
                   + body + "\nreturn 0;}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[0] == ctypes.sizeof(_native.Conv3dDesc)
    for f, off in zip(fields, out[1:]):
        assert getattr(_native.Conv3dDesc, f).offset == off, f


def test_conv1x1_desc_struct_layout(libpath, tmp_path):
    """sizeof/offsetof of estd_conv1x1_desc as the C compiler sees it == the ctypes mirror."""
    from estdepth_amd import _native
    src = tmp_path / "layout1.c"
    names = {"in_": "in"}
    fields = [f[0] for f in _native.Conv1x1Desc._fields_]
    body = "\n".join('printf("%%zu\\n", offsetof(estd_conv1x1_desc, %s));' % names.get(f, f) for f in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "estd_hip.h"\nint main(){printf("%zu\\n", sizeof(estd_conv1x1_desc));\n'
                   + body + "\nreturn 0;}\n")
    exe = tmp_path / "layout1"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[0] == ctypes.sizeof(_native.Conv1x1Desc)
    for f, off in zip(fields, out[1:]):
        assert getattr(_native.Conv1x1Desc, f).offset == off, f
    assert _native.lib().estd_conv1x1_nhwc(None, None) == -1
    assert _native.lib().estd_conv1x1_nhwc(ctypes.byref(_native.Conv1x1Desc()), None) == -1


def test_conv2d_taps_desc_struct_layout(libpath, tmp_path):
    """sizeof/offsetof of estd_conv2d_taps_desc as the C compiler sees it == the ctypes mirror; NULL / empty descriptors are argument errors."""
    from estdepth_amd import _native
    src = tmp_path / "layout2.c"
    names = {"in_": "in"}
    fields = [f[0] for f in _native.Conv2dTapsDesc._fields_]
    body = "\n".join('printf("%%zu\\n", offsetof(estd_conv2d_taps_desc, %s));' % names.get(f, f) for f in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "estd_hip.h"\nint main(){printf("%zu\\n", sizeof(estd_conv2d_taps_desc));\n'
                   + body + "\nreturn 0;}\n")
    exe = tmp_path / "layout2"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[0] == ctypes.sizeof(_native.Conv2dTapsDesc)
    for f, off in zip(fields, out[1:]):
        assert getattr(_native.Conv2dTapsDesc, f).offset == off, f
    assert _native.lib().estd_conv2d_taps_nhwc(None, None) == -1
    assert _native.lib().estd_conv2d_taps_nhwc(ctypes.byref(_native.Conv2dTapsDesc()), None) == -1
    assert _native.lib().estd_stem7x7s2_nhwc(None, None, None, None, None, 1, 8, 8, None) == -1
    assert _native.lib().estd_maxpool3x3s2_nhwc(None, None, 1, 8, 8, 4, None) == -1
    assert _native.lib().estd_avgpool_nhwc(None, None, 1, 8, 8, 4, 2, None) == -1


def test_missing_library_fails_loudly(monkeypatch):
    from estdepth_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libestd_hip.so")
    with pytest.raises(RuntimeError, match="no CPU/eager fallback"):
        _native.lib()


def test_cpu_tensors_are_rejected():
    """The product path has no CPU fallback: CPU tensors raise instead of silently running eager code."""
    import torch
    from estdepth_amd import homo_warping
    with pytest.raises(RuntimeError):
        homo_warping(torch.zeros(1, 4, 8, 8), torch.eye(4)[None], torch.eye(4)[None], torch.ones(1, 4))


def test_default_library_exports_nothing_without_a_default_caller(libpath):
    """The default build carries only kernels with a default caller: the estd_* symbols the library exports are EXACTLY the ones the header
    declares outside #ifdef ESTD_BUILD_AB (the superseded A/B kernels -- depth-only / row-only Winograd, the bf16 operand splits, the
    operand-reuse two-axis kernel -- are compiled, exported, bound and tested only with ESTD_BUILD_AB=1), and every one of them is called
    from the host layer (the ctypes front-end estdepth_amd/ops.py / camera.py or the operator library csrc/torch_ops.cpp)."""
    from estdepth_amd import _native
    if _native.has_ab():
        pytest.skip("library built with ESTD_BUILD_AB=1")
    out = subprocess.check_output(["nm", "-D", "--defined-only", libpath], text=True)
    exported = sorted(set(re.findall(r"\b[TW] (estd_[a-z0-9_]+)$", out, flags=re.M)))
    assert exported == _declared(), (set(exported) ^ set(_declared()))
    assert not [n for n in exported if any(k in n for k in ("_split", "wino2x")) or n.endswith("_wino")], exported
    host = ""
    for rel in ("estdepth_amd/ops.py", "estdepth_amd/camera.py", "estdepth_amd/_native.py", "estdepth_amd/csrc/torch_ops.cpp"):
        host += open(os.path.join(ROOT, rel)).read()
    calls = set(re.findall(r"\b(estd_[a-z0-9_]+)\s*\(", host)) | set(re.findall(r"lib\(\)\.(estd_[a-z0-9_]+)", host))
    meta = {"estd_version"}                 # the ABI version query: for foreign hosts, no kernel behind it
    assert not [n for n in exported if n not in calls | meta], [n for n in exported if n not in calls | meta]
