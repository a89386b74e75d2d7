"""ISA-level invariants of the predict path's kernels, checked on the device assembly of ``engine_predict.hip`` (hipcc cross-compiles
without a GPU; ~10 s).  Round 4 found that a spilled register reloaded behind a store or an atomic costs that operation's round trip
(the memory counter is in order: profiles/r04_experiments.md section 10) -- 4-6 % per kernel.  The fixes are source idioms (lane index
opaque per tile, opaque row limits, requests taken before the closing atomics) that a later edit can undo without any test noticing:
this test notices."""
import os
import re
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def predict_isa(tmp_path_factory):
    from chgnet_amd import build

    try:
        hipcc = build.hipcc_path()
    except RuntimeError:
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "engine_predict.s"
    flags = [f for f in build.HIP_FLAGS if not f.startswith("-W")]
    cmd = [hipcc, *flags, "-w", f"-I{build.INCLUDE}", f"-I{build.CSRC}", "--cuda-device-only", "-S",
           os.path.join(build.CSRC, "engine_predict.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    text = out.read_text()
    starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_ZN\w+):", text, re.M)]
    starts.append((len(text), "END"))
    bodies = {name: text[a:b] for (a, name), (b, _) in zip(starts, starts[1:])}
    meta = {m.group(1): (int(m.group(2)), int(m.group(3)))
            for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text)}
    return bodies, meta


# the kernels of a predict step (names as mangled substrings): tile kernels, row GEMMs, embeddings
HOT = ["k_atomconv_fwd", "k_atomconv_bwdILb0", "k_angleILb1ELb0", "k_angleILb1ELb1", "k_angleILb0ELb0", "k_angleILb0ELb1",
       "k_angle_bwd_wILb1", "k_angle_bwd_wILb0", "k_angleupd_fwd_a", "k_rows_gemm", "k_bond_embed_t", "k_angle_embed_t"]


def test_predict_kernels_do_not_spill(predict_isa):
    bodies, meta = predict_isa
    seen = set()
    for name, (vgprs, spills) in meta.items():
        hit = [h for h in HOT if h in name]
        if not hit:
            continue
        seen.add(hit[0])
        allowed = 2 if "k_angleILb1ELb1" in name else 0      # row-order BondConv adjoint (MD batches): one 64-bit pointer, reloaded mid-phase
        assert spills <= allowed, f"{name}: {spills} spilled registers ({vgprs} VGPRs)"
        assert vgprs <= 256
    assert seen == set(HOT), f"kernels not found in the assembly: {set(HOT) - seen}"


def test_no_spill_reload_sits_behind_a_store_or_atomic(predict_isa):
    """A scratch reload within a few instructions after a global store / atomic, followed by a wait for it, waits for that operation."""
    bodies, _ = predict_isa
    vm = re.compile(r"\s(global_store|global_atomic)")
    for name, body in bodies.items():
        if not any(h in name for h in HOT):
            continue
        lines = body.split("\n")
        for i, line in enumerate(lines):
            if "scratch_load" not in line:
                continue
            recent = [k for k in range(max(0, i - 12), i) if vm.search(lines[k])]
            assert not recent, f"{name}: spill reload at line {i} right behind `{lines[recent[-1]].strip()}`"


def test_atomconv_forward_latch_has_no_wait_behind_its_closing_atomics(predict_isa):
    """The software-pipelined AtomConv forward takes the next tiles' requests before its atomics (gather_take) and has no early
    `continue`: between the last atomic of the loop body and the loop's back edge there is no wait on the vector-memory counter."""
    bodies, _ = predict_isa
    name = next(n for n in bodies if "k_atomconv_fwd" in n)
    lines = bodies[name].split("\n")
    headers = [i for i, l in enumerate(lines) if "Loop Header: Depth=1" in l]
    assert headers
    head = headers[-1]                                   # the tile loop (earlier loops stage the weights)
    before = lines[max(0, head - 60):head]               # the latch block is laid out right before the header
    last_atomic = max((i for i, l in enumerate(before) if "global_atomic_add" in l), default=None)
    assert last_atomic is not None, "latch block without the closing atomics: layout changed, update this test"
    waits = [l.strip() for l in before[last_atomic:] if re.search(r"s_waitcnt\s+vmcnt", l)]
    assert not waits, f"waits behind the closing atomics: {waits}"


# ---- the fine-tuning unit (engine_train.hip): second-order tile kernels ---------------------------------------------------------
@pytest.fixture(scope="module")
def train_isa(tmp_path_factory):
    from chgnet_amd import build

    try:
        hipcc = build.hipcc_path()
    except RuntimeError:
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa_train") / "engine_train.s"
    flags = [f for f in build.HIP_FLAGS if not f.startswith("-W")]
    cmd = [hipcc, *flags, "-w", f"-I{build.INCLUDE}", f"-I{build.CSRC}", "--cuda-device-only", "-S",
           os.path.join(build.CSRC, "engine_train.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    text = out.read_text()
    starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_ZN\w+):", text, re.M)]
    starts.append((len(text), "END"))
    bodies = {name: text[a:b] for (a, name), (b, _) in zip(starts, starts[1:])}
    meta = {m.group(1): (int(m.group(2)), int(m.group(3)))
            for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text)}
    return bodies, meta


# spilled 32-bit registers each second-order tile kernel is allowed (what round 5 left; profiles/r05_experiments.md sections 13, 15: the
# reverse kernels with a hidden layer spilled 107 / 90 before the hidden activation's derivatives were parked in the dump rows, and the
# accumulator-layout forms 225-440 before they were cut to what fits)
T2_SPILL_BUDGET = {"k2_atomILb0E": 0, "k2_atomILb1E": 12, "k2_angleILb0ELb0E": 4, "k2_angleILb1ELb0E": 28, "k2_angleILb0ELb1E": 28,
                   "k2_angleILb1ELb1E": 56}


def test_second_order_tile_kernels_stay_inside_their_spill_budget(train_isa):
    _, meta = train_isa
    seen = set()
    for name, (vgprs, spills) in meta.items():
        for key, budget in T2_SPILL_BUDGET.items():
            if key in name:
                seen.add(key)
                assert spills <= budget, f"{name}: {spills} spilled registers (budget {budget})"
                assert vgprs <= 256
    assert seen == set(T2_SPILL_BUDGET), f"kernels not found in the assembly: {set(T2_SPILL_BUDGET) - seen}"
    # the first-order fine-tuning adjoints and the frequency-gradient tile kernels do not spill at all
    for name, (vgprs, spills) in meta.items():
        if any(k in name for k in ("k_atomconv_bwdILb1", "k2_freq_grad_t", "k2_angle_freq_grad_t", "k_xty3")):
            assert spills == 0, f"{name}: {spills} spilled registers"


def test_layernorm_affine_sums_are_not_a_read_modify_write_chain_through_scratch(train_isa):
    """The four running LayerNorm-affine sums live across a whole second-order kernel.  Summed row by row, an allocator that keeps them
    in scratch turns every row into a dependent store -> load round trip (found on the GPU: 13.9 instead of 8.3 ms for the AngleUpdate
    reverse kernel); the accumulator-layout kernels sum into locals first.  In their assembly no scratch store is followed by a scratch
    load of the same offset within a few instructions more than a handful of times."""
    bodies, _ = train_isa
    off = re.compile(r"scratch_(load|store)_dword\w*\s+.*?offset:(\d+)")
    for key in ("k2_angleILb0ELb1E", "k2_angleILb0ELb0E", "k2_angleILb1ELb0E", "k2_atomILb0E"):
        name = next(n for n in bodies if key in n)
        lines = [l for l in bodies[name].split("\n") if "scratch_" in l or l.strip().startswith(("v_", "ds_", "global_"))]
        chains = 0
        for i, line in enumerate(lines):
            m = off.search(line)
            if not m or m.group(1) != "store":
                continue
            for later in lines[i + 1:i + 6]:
                m2 = off.search(later)
                if m2 and m2.group(1) == "load" and m2.group(2) == m.group(2):
                    chains += 1
                    break
        assert chains <= 4, f"{name}: {chains} store -> load round trips through the same scratch slot"
