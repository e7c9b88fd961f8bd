"""CPU-side checks of the C-ABI boundary: libcvk.so loads, exports exactly what include/cvk.h declares, and the
Python binding table matches.  No compute calls (no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "cvk.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cvk_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_expected_surface():
    syms = header_symbols()
    assert "cvk_conv3x3_fwd" in syms and "cvk_conv3x3_wgrad" in syms and "cvk_softmax_ce_fwd" in syms
    assert len(syms) >= 30


def test_library_builds_loads_and_exports_every_declared_symbol():
    import ctypes
    from pytorch_camvid_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = _lib.load()
    assert lib.cvk_version() == 100
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in header_symbols():
        assert hasattr(raw, s), f"{s} declared in include/cvk.h but not exported by libcvk.so"
    assert sorted(_lib.SIGNATURES) == header_symbols(), "Python binding table out of sync with include/cvk.h"


def test_stale_library_is_refused(tmp_path):
    """The library carries the hash of the header it was compiled against; load() compares it with the header it sees.  A library
    built from any other header (same symbol names, other argument lists) must not load."""
    import ctypes
    from pytorch_camvid_amd import _lib
    lib = _lib.load()
    assert lib.cvk_abi_hash() == _lib.header_abi_hash()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    _lib.check_abi(raw)                                   # the real pair passes
    other = tmp_path / "cvk.h"
    other.write_bytes(open(_lib.HEADER, "rb").read() + b"\n/* one more argument somewhere */\n")
    with pytest.raises(_lib.CvkError, match="another include/cvk.h"):
        _lib.check_abi(raw, header=str(other))

    class NoStamp:                                        # a library from before the stamp existed
        pass
    with pytest.raises(_lib.CvkError, match="predates the ABI stamp"):
        _lib.check_abi(NoStamp())


def test_integration_doc_matches_the_header():
    """INTEGRATION.md is the binding guide a maintainer reads: every cvk_* name it mentions must exist in include/cvk.h, every
    entry point of the header must be mentioned, and the stated counts (entry points, .hip files) must be the real ones."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    syms = set(header_symbols())
    mentioned = set(re.findall(r"\b(cvk_[a-z0-9_]+)\b", doc)) - {"cvk_view"}          # cvk_view is the struct type
    assert not (mentioned - syms), f"INTEGRATION.md names entry points that do not exist: {sorted(mentioned - syms)}"
    assert not (syms - mentioned), f"entry points missing from INTEGRATION.md: {sorted(syms - mentioned)}"
    m = re.search(r"declares (\d+) `extern \"C\"` entry points", doc)
    assert m and int(m.group(1)) == len(syms), (m and m.group(1), len(syms))
    words = {8: "eight", 9: "nine", 10: "ten", 11: "eleven", 12: "twelve", 13: "thirteen", 14: "fourteen", 15: "fifteen"}
    nhip = len([f for f in os.listdir(os.path.join(ROOT, "pytorch-camvid_amd", "csrc")) if f.endswith(".hip")])
    assert f"{words[nhip]} `.hip` files" in doc, nhip
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    m = re.search(r"`include/cvk.h` \((\d+) entry points\)", design)
    assert m and int(m.group(1)) == len(syms), (m and m.group(1), len(syms))


def test_argument_validation_without_gpu():
    """Argument errors are detected before any launch, return CVK_EINVAL and set the error string."""
    from pytorch_camvid_amd import _lib
    lib = _lib.load()
    rc = lib.cvk_conv3x3_fwd(None, None, None, None, None, 1, 8, 8, 4, 8, 8, None)
    assert rc == -1 and b"null" in lib.cvk_last_error_string()
    rc = lib.cvk_conv3x3_fwd(16, 16, None, 16, None, 1, 8, 8, 3, 8, 8, None)
    assert rc == -1 and b"multiple of 4" in lib.cvk_last_error_string()
    assert lib.cvk_conv3x3_wgrad_workspace_bytes(8, 360, 480, 64, 64) > 0
    assert lib.cvk_bn_bwd_blocks(1382400) == 512 and lib.cvk_ce_blocks(1025) == 2


def test_module_surface_matches_reference_and_fails_loudly_on_cpu():
    import torch
    import pytorch_camvid_amd as A
    net = A.get_model("unet", 3, 12)
    assert len(net.state_dict()) == 161 and sum(p.numel() for p in net.parameters()) == 34533924
    seg = A.get_model("segnet", 3, 12)
    assert len(seg.state_dict()) == 182 and sum(p.numel() for p in seg.parameters()) == 29449956
    with pytest.raises(ValueError):
        A.get_model("nope", 3, 12)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 3, 32, 32))
    # checkpoints interchange with the reference layout: plain OIHW tensors load into channels_last storage
    from oracle import torch_ref as R
    torch.manual_seed(3)
    ref = R.build("unet", 3, 12)
    net.load_state_dict(ref.state_dict())
    for (k, a), (_, b) in zip(net.state_dict().items(), ref.state_dict().items()):
        assert torch.equal(a, b), k
    ref.load_state_dict(net.state_dict())


def test_product_library_reads_no_environment():
    """a stray variable in a user's environment must not be able to change a kernel — least of all select one of the
    wrong-result ablation variants the timing experiments use.  Those (and every CVK_* switch of the C side) are compiled only with
    -DCVK_EXPERIMENTS into lib/libcvk_exp.so (`make experiments`, loaded by tools/ through CVK_LIB_PATH).  The product library must
    not import getenv and must not contain a CVK_* name, a *_DBG name in particular."""
    import subprocess
    from pytorch_camvid_amd import _lib
    path = os.path.join(ROOT, "pytorch-camvid_amd", "lib", "libcvk.so")
    if not os.path.exists(path):
        _lib.build()
    data = open(path, "rb").read()
    names = sorted(set(m.decode() for m in re.findall(rb"CVK_[A-Z0-9_]{3,}", data)))
    assert not [n for n in names if n.endswith("_DBG")], names
    assert not names, f"environment-style names inside the product library: {names}"
    undefined = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined, "libcvk.so imports getenv"
    # and the superseded kernel generations are gone from the default build (they live in git history)
    assert b"k_conv_bf16p" not in data and b"k_wgrad_bf16sI" not in data
    # the product never loads the experiments build by itself: only an explicit CVK_LIB_PATH does
    assert _lib.LIB_PATH == (os.environ.get("CVK_LIB_PATH") or path)
