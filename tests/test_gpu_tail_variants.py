"""The two implementations of the tail stages (stages 3/4 of msd.hpp's cascade) against each other and the oracle:
k_tail_s (compile-time taps, every chain of sdrx.cpp:1588-1668) and k_tail (generic, run-time taps)."""
import numpy as np
import pytest

from conftest import RTOL, rel_err_rows
from sdrx_amd import bank as sb, chains, synth

pytestmark = pytest.mark.gpu

WHOLE = 0xFFFFFFFF      # block_hint: a whole segment per lane (the layout of a large bank)


def _run(generic, rate, offs, blocks, hint, fused=sb.FUSED_OFF):
    ch = chains.chain(rate)
    bank = sb.ChannelBank(sb.translators_for(ch, offs), ch.stage_tuples(), block_hint=hint,
                          tuning={"tail_generic": int(generic), "fused_tail": fused})
    iq = []
    for x in blocks:
        iq.append(bank.process(x, want_audio=False)[0])
    bank.close()
    return np.concatenate(iq, axis=1)


@pytest.mark.parametrize("rate", ["10", "2.4", "0.96", "1.2", "1.44", "1.6", "1.92", "6", "2.56"])
def test_specialised_and_generic_tail_kernels_match_the_oracle(oracle, rate):
    ch = chains.chain(rate)
    offs = [0, 7, -11, 3, 1]
    L = ch.block_len
    stream = synth.synth_iq(17, 0, 2 * L + 1234, 0.25, [sb.translators_for(ch, [7])[0]], [0.2], ch.N, 5000)
    cuts = [0, L, L + 999, 2 * L + 1234]                       # a whole block, a short call, the rest
    blocks = [stream[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    ref = np.stack([oracle.msd(t, ch.stage_tuples()).decimate(stream) for t in sb.translators_for(ch, offs)])
    outs = {}
    for generic in (False, True):
        for hint in (0, WHOLE):                                # auto layout (P > 1 for a small bank) and P == 1
            iq = _run(generic, rate, offs, blocks, hint)
            assert iq.shape == ref.shape
            assert rel_err_rows(iq, ref) <= RTOL, (rate, generic, hint)
            outs[(generic, hint)] = iq
    # the layout of the stage-2 scratch never changes a result
    for generic in (False, True):
        assert np.array_equal(outs[(generic, 0)].view(np.uint32), outs[(generic, WHOLE)].view(np.uint32))
    # the two kernels add the taps of the last stage in a different order: close, not identical
    assert rel_err_rows(outs[(False, 0)], outs[(True, 0)]) <= 2e-6


def test_specialised_tail_many_channels_and_tiny_calls(oracle):
    """300 channels (not a multiple of 8 or 16), calls from 0 to several blocks' worth of samples."""
    ch = chains.chain("10")
    offs = chains.offsets_cycled(300, ch.N)
    L = ch.block_len
    stream = synth.synth_iq(23, 0, 2 * L, 0.25)
    lens = [0, 1, 624, 625, 5000, L - 6250, L, 10]
    lens.append(2 * L - sum(lens))
    blocks, p = [], 0
    for n in lens:
        blocks.append(stream[p:p + n])
        p += n
    a0 = _run(False, "10", offs, blocks, WHOLE)
    a1 = _run(True, "10", offs, blocks, WHOLE)
    assert rel_err_rows(a0, a1) <= 2e-6
    for c in (0, 150, 299):
        y = oracle.msd(sb.translators_for(ch, [offs[c]])[0], ch.stage_tuples()).decimate(stream)
        assert rel_err_rows(a0[c:c + 1], y[None]) <= RTOL
