"""The BER harness (labrador_ldpc_amd.perftest, GPU counterpart of perftest/src/main.rs): the reference
records no curves, so the checks are structural -- BER falls monotonically with SNR, matches the CPU
oracle's BER on the same noise convention within sampling error, and the CSV has the reference's columns."""
import numpy as np
import pytest

import oracle
from labrador_ldpc_amd import LDPCCode
from labrador_ldpc_amd import perftest

pytestmark = pytest.mark.gpu


def test_ber_is_monotone_and_matches_oracle():
    code = LDPCCode.TC128
    bers = []
    for snr in (1.0, 3.0, 5.0):
        trials, bits, errors, ber, fe = perftest.ms_trials(code, snr, "ebn0", maxiters=50, batch=16384,
                                                           max_bits=2e6, max_errors=10**9)
        assert bits == trials * code.k() and trials >= 16384
        bers.append(ber)
    assert bers[0] > bers[1] > bers[2]
    # oracle at the middle point, same convention, independent noise
    rng = np.random.default_rng(4)
    llrs, cws = oracle.awgn_llrs(code, rng, 4096, 3.0, np.float32)
    out, _, _, _ = oracle.decode_ms_batch(code, llrs, 50)
    err = np.unpackbits(out[:, : code.k() // 8] ^ cws[:, : code.k() // 8], axis=1).sum()
    ber_cpu = err / (4096 * code.k())
    assert abs(bers[1] - ber_cpu) < 0.35 * ber_cpu + 1e-4


def test_cli_prints_reference_csv_columns(capsys):
    perftest.main(["--code", "TC128", "--snrs", "2.0", "--noise", "perftest", "--maxiters", "20",
                   "--batch", "4096", "--max-bits", "1e5"])
    line = capsys.readouterr().out.strip().splitlines()[-1].split(",")
    assert line[0] == "TC128" and line[1] == "2.00" and len(line) == 6
    assert int(line[3]) == int(line[2]) * 64 and float(line[5]) > 0
