"""Import FIRST in a kernel-experiment script: makes the package load lib/libcvk_exp.so — the same sources compiled with
-DCVK_EXPERIMENTS (`make -C pytorch-camvid_amd/csrc experiments`): environment knobs (CVK_BF16P*, CVK_STREAM_HINTS, CVK_WGRAD_PRIO,
CVK_W2D_NO_STAGGER, ...), in-kernel time stamps and the ablation variants with WRONG results (CVK_BF16H_DBG, CVK_WGRAD_DBG).  The
product library lib/libcvk.so contains none of them and never reads the environment (tests/test_abi.py)."""
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CSRC = os.path.join(_ROOT, "pytorch-camvid_amd", "csrc")
EXP_LIB = os.path.join(_ROOT, "pytorch-camvid_amd", "lib", "libcvk_exp.so")

if "CVK_LIB_PATH" not in os.environ:
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".h"))]
    if not os.path.exists(EXP_LIB) or os.path.getmtime(EXP_LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _CSRC, "-j8", "experiments"], stdout=subprocess.DEVNULL)
    os.environ["CVK_LIB_PATH"] = EXP_LIB
