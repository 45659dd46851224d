"""Build-time checks on the search kernels' ISA (no GPU: hipcc cross-compiles gfx950 here).

Round 5 found three places where the compiler's conservative `s_waitcnt vmcnt(0)` -- one counter covers loads AND stores on
gfx9, and it cannot be counted across branches -- put a memory round trip on every round's dependent chain
(profiles/r05_s_stage_waits.txt).  The fixes are orderings in the source that a later edit can undo without any test turning red
(results stay bit-identical), so the properties are pinned where they live: in the generated code.

  1. k_search_mlp's rounds: no wait on the vector-memory counter follows a twisted-word store of the word staging before the
     next load is issued (stage_finish stores last: SMZ_STAGE_STORES_LAST).
  2. the same rounds: the six source-word loads of the NEXT round are issued in the selection, together with the parent-row
     loads (SMZ_EARLY_STAGE), i.e. before the first s_setprio that opens the network phase.
  3. k_search_vision's rounds: no 16-byte global load at all -- the tower biases come from LDS (SMZ_VISION_BIAS_LDS).
  4. trees in global memory: the blocks of two selection passes are requested before anything waits (SMZ_SELECT_TWO_PASSES).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stochastic-muzero_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# the library's own flags (csrc/Makefile)
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wno-unused-function", "-Wno-unused-variable",
         "-Wno-unused-const-variable", "-S", "--cuda-device-only"]

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("make") is None or os.environ.get("SMZ_SKIP_ISA_TESTS"),
                                reason="needs hipcc (cross-compiles without a GPU)")


def _isa(tmp_path, source, extra=()):
    out = tmp_path / (source + ".s")
    r = subprocess.run([HIPCC, *FLAGS, *extra, "-o", str(out), source], cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text().split("\n")


def _kernel(lines, mangled_prefix):
    """The instructions of one kernel, each with whether its basic block belongs to the kernel's MAIN loop -- the simulation
    rounds: the outermost loop with the most basic blocks -- or to a loop nested in it (the assembler's comments name every
    block's loop header and every inner header's parent)."""
    start = next(i for i, l in enumerate(lines) if l.startswith(mangled_prefix) and l.rstrip().split(";")[0].strip().endswith(":"))
    blocks, parent, cur = [], {}, None           # (header of the block's loop or None, [instructions])
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        label = re.match(r"^\.L(BB\d+_\d+):", l)
        if label or l.startswith("; %bb."):
            h = re.search(r"in Loop: Header=(BB\d+_\d+)", l)
            p = re.search(r"Parent Loop (BB\d+_\d+)", l)
            if label and (p or "Loop Header" in l):   # a loop's header block belongs to its own loop (an inner one names its parent)
                if p:
                    parent[label.group(1)] = p.group(1)
                cur = label.group(1)
            else:
                cur = h.group(1) if h else None
            blocks.append((cur, []))
            continue
        t = l.strip()
        if t and not t.startswith((";", ".")) and blocks:
            blocks[-1][1].append(t)
    def outermost(h):
        while h in parent:
            h = parent[h]
        return h
    count = {}
    for h, ins in blocks:
        if h is not None:
            count[outermost(h)] = count.get(outermost(h), 0) + 1
    main = max(count, key=count.get)
    body = [(t, h is not None and outermost(h) == main) for h, ins in blocks for t in ins]
    assert len(body) > 1000 and sum(1 for _, m in body if m) > 500, "kernel body / main loop not found"
    return body


@pytest.fixture(scope="module")
def part2_isa(tmp_path_factory):
    return _isa(tmp_path_factory.mktemp("isa"), "smz_kernels.hip", ["-DSMZ_PART=2"])


@pytest.fixture(scope="module")
def search_isa(part2_isa):
    # k_search_mlp<2, 2, 1, false, true, false, false, true>: the headline workload's instantiation
    return _kernel(part2_isa, "_ZN12_GLOBAL__N_112k_search_mlpILi2ELi2ELi1ELb0ELb1ELb0ELb0ELb1EEE")


def test_two_selection_passes_are_requested_together_on_global_memory_trees(part2_isa):
    """k_search_mlp<2, 2, 1, false, true, false, false, false> (C5's per-rank shape: trees in global memory): the blocks of two
    passes of the block-parallel selection -- per block three 16-byte loads, the auxiliary pair, the root's two float64 priors --
    are all requested before the first wait on the vector-memory counter (SMZ_SELECT_TWO_PASSES, select_block_request)."""
    body = _kernel(part2_isa, "_ZN12_GLOBAL__N_112k_search_mlpILi2ELi2ELi1ELb0ELb1ELb0ELb0ELb0EEE")
    best, wide, cur, curw = 0, 0, 0, 0
    for t, loop in body:
        if not loop:
            continue
        if t.startswith("global_load_dword"):
            cur += 1
            curw += t.startswith("global_load_dwordx4")
        elif t.startswith(("s_waitcnt vmcnt", "global_store")):
            if curw > wide:
                best, wide = cur, curw
            cur = curw = 0
    # (one pass' block: two 16-byte + an 8-byte load of its twelve words -- the last two are not read --, the auxiliary pair, the
    #  root's priors: 4-5 loads, 2-3 of them 16-byte)
    assert wide >= 4 and best >= 8, f"longest run of block requests without a wait: {best} loads, {wide} of them 16-byte (two passes: >= 8 / >= 4)"


def test_no_wait_behind_a_twisted_word_store(search_isa):
    body = search_isa
    twisted = 0
    for i, (t, loop) in enumerate(body):
        if not (loop and t.startswith("global_store_dword ")):
            continue
        # a twisted word: v_bitop3 (x & 1 ? 0x9908b0df : 0) ^ ... a few instructions earlier
        if not any("bitop3:0x6c" in u or "0x9908b0df" in u for u, _ in body[max(0, i - 14):i]):
            continue
        twisted += 1
        for u, _ in body[i + 1:i + 60]:
            if u.startswith(("global_load", "global_store", "s_endpgm")):
                break                                  # the next request is out before anything waits: fine
            assert "vmcnt" not in u, f"a wait on the vector-memory counter follows a twisted-word store: {u!r} after {t!r}"
    assert twisted >= 2, "stage_finish's stores not found in the round (pattern changed?)"


def test_source_words_are_requested_in_the_selection(search_isa):
    body = search_isa
    prio = [i for i, (t, loop) in enumerate(body) if loop and t.startswith("s_setprio")]
    assert len(prio) >= 2, "the round's two s_setprio not found"
    tree_phase = body[prio[0]:prio[1]]                 # s_setprio 0 (tree phases) ... s_setprio 3 (network phase)
    # the block-parallel selection's early requests: two parent rows + six source words, one basic-block run
    runs, cur = [], 0
    for t, _ in tree_phase:
        if t.startswith("global_load_dword "):
            cur += 1
        elif t.startswith(("s_waitcnt vmcnt", "global_store")):
            runs.append(cur)
            cur = 0
    runs.append(cur)
    assert max(runs) >= 8, f"parent rows + next round's source words are not requested together in the selection (runs of loads: {runs})"


def test_vision_round_reads_no_bias_from_global_memory(tmp_path):
    lines = _isa(tmp_path, "smz_vision_search.hip")
    body = _kernel(lines, "_ZN12_GLOBAL__N_115k_search_visionILi2ELb1ELb0EEE")
    wide = [t for t, loop in body if loop and t.startswith("global_load_dwordx4")]
    assert not wide, f"16-byte global loads inside k_search_vision's rounds (tower biases should come from LDS): {wide[:3]}"
    assert len([t for t, loop in body if loop and t.startswith("ds_read_b128")]) > 20
