"""tools/repro/wwm_lost_update.py, the assembly checker that located round 3's miscompile: it must flag a reload that loses
lanes written since the last store to the same slot -- on a path, not only in straight-line code -- and nothing else."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LOOP = """
kernel_a:
\ts_mov_b64 s[0:1], 0
\tv_writelane_b32 v127, s4, 32
\ts_or_saveexec_b64 s[100:101], -1
\tscratch_store_dword off, v127, off offset:904 ; 4-byte Folded Spill
\ts_mov_b64 exec, s[100:101]
.LBB0_1:
\ts_cbranch_scc1 .LBB0_3
.LBB0_2:
\ts_or_saveexec_b64 s[100:101], -1
\tscratch_store_dword off, v127, off offset:904 ; 4-byte Folded Spill
\ts_mov_b64 exec, s[100:101]
\ts_swappc_b64 s[30:31], s[0:1]
.LBB0_3:
\ts_or_saveexec_b64 s[100:101], -1
\tscratch_load_dword v127, off, off offset:904 ; 4-byte Folded Reload
\ts_mov_b64 exec, s[100:101]
\tv_readlane_b32 s2, v127, 32
\tv_writelane_b32 v127, s2, 32
%s\ts_cbranch_execnz .LBB0_1
.LBB0_4:
\ts_or_saveexec_b64 s[100:101], -1
\tscratch_load_dword v127, off, off offset:16 ; restores the caller's value: another slot
\ts_mov_b64 exec, s[100:101]
\ts_endpgm
.Lfunc_end0:
"""
STORE = "\ts_or_saveexec_b64 s[100:101], -1\n\tscratch_store_dword off, v127, off offset:904 ; 4-byte Folded Spill\n\ts_mov_b64 exec, s[100:101]\n"


def run(tmp_path, text):
    f = tmp_path / "a.s"
    f.write_text(text)
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "repro", "wwm_lost_update.py"), str(f)], capture_output=True, text=True, check=True).stdout


def test_checker_flags_the_latch_that_does_not_store(tmp_path):
    out = run(tmp_path, LOOP % "")           # the latch rewrites lane 32 and loops: the call-free trip (.LBB0_1 -> .LBB0_3) reloads the old value
    assert "reloads with unsaved lanes on some path: 1" in out and "lanes [32]" in out, out


def test_checker_accepts_the_latch_that_stores(tmp_path):
    out = run(tmp_path, LOOP % STORE)
    assert "reloads with unsaved lanes on some path: 0" in out, out
