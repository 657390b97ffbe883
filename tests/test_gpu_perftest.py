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


def test_native_perftest_binary_matches_the_python_harness():
    """harness/perftest.cpp -- the reference's perftest (perftest/src/main.rs) as a native program over the C ABI,
    one worker thread per GPU: CSV columns and stopping rule of the reference, BER statistically equal to the
    Python harness's at the same operating points, monotone in the SNR, and with a device list."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "harness"), "-s"])
    exe = os.path.join(root, "build", "perftest")
    r = subprocess.run([exe, "TM1280", "--noise", "ebn0", "--snrs", "2.0,3.0,3.6,5.0", "--max-bits", "6e7", "--batch", "16384",
                        "--maxiters", "50", "--devices", "0,0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    rows = [ln.split(",") for ln in r.stdout.strip().splitlines()]
    assert [row[0] for row in rows] == ["TM1280"] * 4 and [row[1] for row in rows] == ["2.00", "3.00", "3.60", "5.00"]
    code = LDPCCode.TM1280
    bers = []
    for name, snr, trials, bits, errors, ber in rows:
        assert int(bits) == int(trials) * code.k() and int(errors) >= 1
        assert abs(float(ber) - int(errors) / int(bits)) <= 1e-4 * float(ber)
        assert int(bits) > 6e7 or int(errors) > 5000                              # the reference's stopping rule (:50)
        bers.append(float(ber))
    assert bers[0] > bers[1] > bers[2] > bers[3]
    for snr, ber in ((2.0, bers[0]), (3.0, bers[1])):                              # points with thousands of errors
        _, _, errors, ber_py, _ = perftest.ms_trials(code, snr, "ebn0", maxiters=50, batch=16384, max_bits=6e7)
        assert errors > 1000 and 0.8 < ber / ber_py < 1.25, (snr, ber, ber_py)
