"""The wait states that the kernels' inline assembly relies on, checked on the BUILT code object (tools/hazard_lint.py): every
`global_store_dwordx4 ... nt` has two wait states before a vector instruction overwrites its data registers, every inline
`v_pk_add_f32` behind the f32 MFMAs reads an accumulator no sooner than 12 wait states after the MFMA that wrote it -- over the
control-flow graph of each kernel, as LLVM's hazard recogniser counts them.  Runs on the CPU box (llvm-objdump of the gfx950 code
object inside libsdrxgpu.so); __graft_entry__.build() runs the same lint.  The lint's own logic is pinned on small hand-made
listings: the hazards it must find (round 5's bug: the streaming store without its s_nop) and the cases it must let pass."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import hazard_lint as hl  # noqa: E402

LIB = os.path.join(ROOT, "sdrx_amd", "lib", "libsdrxgpu.so")


def _listing(body):
    lines = ["0000000000001000 <k>:"]
    addr = 0x1000
    for ins in body:
        # (llvm-objdump prints a branch's target behind the encoding: "s_cbranch_scc1 25   // 00000005BFBC: BF850019 <k+0x2b24>")
        ins, _, tgt = ins.partition(" <")
        lines.append(f"\t{ins:58s} // {addr:012X}: 00000000" + (" <" + tgt if tgt else ""))
        addr += 8
    return "\n".join(lines) + "\n"


def _lint(body):
    funcs = hl.parse(_listing(body))
    return hl.lint_function("k", funcs["k"])[0]


PK = "v_pk_add_f32 v[10:11], v[0:1], v[4:5] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
MF = "v_mfma_f32_16x16x4_f32 v[0:3], v20, v21, v[0:3]"


def test_mfma_result_read_too_early_is_found_and_enough_wait_states_pass():
    assert len(_lint([MF, "s_nop 3", PK])) == 1                               # 4 wait states
    assert len(_lint([MF] + ["ds_read_b64 v[30:31], v40"] * 11 + [PK])) == 1  # 11 instructions in between
    assert _lint([MF] + ["ds_read_b64 v[30:31], v40"] * 12 + [PK]) == []
    assert _lint([MF, "s_nop 7", "s_nop 3", PK]) == []                       # 8 + 4
    # another MFMA that writes other registers does not count as the writer
    assert _lint([MF, "s_nop 7", "s_nop 3", "v_mfma_f32_16x16x4_f32 v[40:43], v20, v21, v[40:43]", PK]) == []


def test_wait_states_only_count_if_they_lie_on_every_path():
    # a forward branch around the nops: on the taken path the read follows the MFMA at once
    body = [MF, "s_cbranch_scc1 2 <k+0x20>", "s_nop 7", "s_nop 3", PK]
    assert len(_lint(body)) == 1
    # a loop whose back edge arrives right in front of the read, with the MFMA at the loop's end
    body = ["s_nop 7", "s_nop 7", PK, "v_add_f32 v50, v51, v52", MF, "s_cbranch_scc1 65531 <k+0x10>", "s_endpgm"]
    assert len(_lint(body)) == 1


def test_streaming_store_needs_two_wait_states_before_its_data_is_overwritten():
    st = "global_store_dwordx4 v48, v[24:27], s[34:35] nt"
    assert len(_lint([st, "v_pk_add_f32 v[24:25], v[0:1], v[4:5]"])) == 1          # round 5's bug
    assert len(_lint([st, "s_nop 0", "v_mov_b32 v26, v1"])) == 1
    assert _lint([st, "s_nop 1", "v_pk_add_f32 v[24:25], v[0:1], v[4:5]"]) == []
    assert _lint([st, "v_add_u32 v48, s23, v112", "v_add_u32 v49, s23, v112", "v_mov_b32 v24, v1"]) == []
    assert _lint([st, "v_mov_b32 v60, v1", "v_mov_b32 v61, v1"]) == []                # other registers
    assert _lint(["global_store_dwordx4 v48, v[24:27], s[34:35]", "v_mov_b32 v24, v1"]) == []   # the compiler's own stores are its to space


@pytest.mark.skipif(not os.path.exists(LIB), reason="libsdrxgpu.so not built")
def test_the_built_library_is_clean_and_the_lint_sees_what_it_is_there_for():
    bad, stats = hl.lint(LIB)
    assert stats["inline_pk_add"] >= 240 and stats["nt_stores"] >= 24, stats      # the patterns still match the build
    assert bad == [], "\n".join(bad[:10])
