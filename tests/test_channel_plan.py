"""Channel names -> translator offsets: sdrxgpu_parse_channel / _tuner_fq / _channel_to_offset /
_channels_fit_bandwidth of the C ABI against (1) answers worked out by hand from the reference's text
(src/sdrx.cpp:957-1021, 1398-1408, 1424-1478) and (2) the oracle's independent restatement over every channel of the
airband.  Integer work: exact.  (sdrx.cpp cannot be compiled here -- popt / ALSA / FFTW -- so no reference run backs
these; the known answers below follow the reference's own examples, doc/USING.md:82-101.)"""
import itertools

import pytest

from sdrx_amd import bank as sb
from sdrx_amd import chains

SUB = {"00": (0, 0), "05": (0, 0), "10": (1, 8333), "15": (2, 16667), "25": (3, 25000), "30": (3, 25000),
       "35": (4, 33333), "40": (5, 41667), "50": (6, 50000), "55": (6, 50000), "60": (7, 58333), "65": (8, 66667),
       "75": (9, 75000), "80": (9, 75000), "85": (10, 83333), "90": (11, 91667)}


def all_channels(lo_mhz=118, hi_mhz=137):
    return [f"{m}.{t}{s}" for m in range(lo_mhz, hi_mhz) for t in range(10) for s in SUB]


def test_known_answers_by_hand():
    # doc/USING.md:82: ./sdrx 118.105 118.280 118.405 118.505 -> lowest 118.100 MHz, highest 118.500 MHz, midpoint 118.3 MHz
    names = ["118.105", "118.280", "118.405", "118.505"]
    assert [sb.parse_channel(n) for n in names] == [118100000, 118275000, 118400000, 118500000]
    assert sb.tuner_fq(names) == 118300000
    # 118.105: base 118.1 MHz = -2 x 100 kHz = -24 steps, "05" = step 0;  118.280: base 118.2 = -12, "80" = step 9
    assert [sb.channel_to_offset(n, 118300000) for n in names] == [-24, -3, 12, 24]
    # one 100 kHz band, both schemes on one grid of 12 steps: 8.33 kHz names 005 010 015 030 035 040 055 060 065 080 085 090,
    # 25 kHz names 000 025 050 075 (= steps 0 3 6 9, the same chain: SURVEY 8 a3)
    band = ["118.300", "118.305", "118.310", "118.315", "118.325", "118.330", "118.335", "118.340", "118.350", "118.355",
            "118.360", "118.365", "118.375", "118.380", "118.385", "118.390"]
    assert [sb.channel_to_offset(n, 118300000) for n in band] == [0, 0, 1, 2, 3, 3, 4, 5, 6, 6, 7, 8, 9, 9, 10, 11]
    assert [sb.parse_channel(n) - 118300000 for n in band] == [0, 0, 8333, 16667, 25000, 25000, 33333, 41667, 50000, 50000,
                                                               58333, 66667, 75000, 75000, 83333, 91667]
    # below the tuner: (fq_diff / 100000) * 12 with fq_diff a negative multiple of 100 kHz, plus the step inside the band
    assert sb.channel_to_offset("117.990", 118300000) == -4 * 12 + 11
    # the tuner is rounded to the nearest 100 kHz: one channel alone sits up to 11 steps above it
    assert sb.tuner_fq(["118.190"]) == 118200000 and sb.channel_to_offset("118.190", 118200000) == -1
    assert sb.tuner_fq(["118.140"]) == 118100000 and sb.channel_to_offset("118.140", 118100000) == 5
    # 80 % rule at 10 MS/s: 8 MHz of span (sdrx.cpp:1437); these are the +-480 steps SURVEY quotes
    assert sb.channels_fit_bandwidth(["118.000", "126.000"], 10_000_000)
    assert not sb.channels_fit_bandwidth(["118.000", "126.010"], 10_000_000)
    t = sb.tuner_fq(["118.000", "126.000"])
    assert t == 122000000 and sb.channel_to_offset("118.000", t) == -480 and sb.channel_to_offset("126.000", t) == 480
    # 2.4 MS/s: 1.92 MHz
    assert sb.channels_fit_bandwidth(["118.000", "119.915"], 2_400_000) and not sb.channels_fit_bandwidth(["118.000", "119.925"], 2_400_000)


def test_invalid_names_are_refused():
    for bad in ["118105", "118.10", "118.1055", "1.105", "11811.105", "118.1a5", "118.120", "118.145", "118.170", "118.195", "", "abc.def", "4000.000"]:
        with pytest.raises(sb.SdrxGpuError) as e:
            sb.parse_channel(bad)
        assert e.value.code == -1
    with pytest.raises(sb.SdrxGpuError):
        sb.channel_to_offset("118.120", 118000000)
    with pytest.raises(sb.SdrxGpuError):
        sb.tuner_fq([])


def test_library_equals_oracle_over_the_airband(oracle):
    names = all_channels()
    assert len(names) == 19 * 10 * 16
    for n in names:
        assert sb.parse_channel(n) == oracle.parse_fq(n, True) != 0
    for tuner in (118000000, 127500000, 136900000, 100000000):
        for n in names[::7]:
            assert sb.channel_to_offset(n, tuner) == oracle.channel_to_offset(n, tuner)
    # offsets are consistent with the frequencies: offset * 8333.33 Hz == channel - tuner to within the grid's rounding
    for n in names[::11]:
        off = sb.channel_to_offset(n, 127500000)
        assert abs(off * 100000 / 12 - (sb.parse_channel(n) - 127500000)) < 1.0
    import random
    rnd = random.Random(4)
    for _ in range(300):
        pick = rnd.sample(names, rnd.randint(1, 6))
        assert sb.tuner_fq(pick) == oracle.tuner_fq(pick)
        for rate in (960000, 2400000, 6000000, 10000000):
            assert sb.channels_fit_bandwidth(pick, rate) == oracle.verify_bandwidth(pick, rate)
    # oracle's non-aeronautical parse_fq (frequency notation) on a few strings, by hand
    assert oracle.parse_fq("118.105", False) == 118105000 and oracle.parse_fq("433.92", False) == 433920000
    assert oracle.parse_fq("118", False) == 0 and oracle.parse_fq("1.5", False) == 0


def test_offsets_feed_the_translator_and_the_bank_setup(oracle):
    """Names -> tuner -> offsets -> tables: the set-up loop of sdrx.cpp:1680-1690 end to end on the host."""
    names = ["118.105", "118.280", "118.300", "118.405", "118.505"]
    ch = chains.chain("2.4")
    t = sb.tuner_fq(names)
    offs = [sb.channel_to_offset(n, t) for n in names]
    assert offs == [-24, -3, 0, 12, 24]
    tabs = sb.translators_for(ch, offs)
    assert tabs[2] is None                                   # at the tuner centre: empty translator, sdrx.cpp:1685
    for o, tab in zip(offs, tabs):
        if o:
            ref = oracle.make_translator(ch.N, ch.z, o)
            assert (tab.view("u4") == ref.view("u4")).all()
