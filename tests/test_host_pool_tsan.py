"""GpuChannelPool's host logic (asr-decoder_amd/host/wfst-host.cc) under ThreadSanitizer, without a device: the pool, its batcher thread
and N x GpuLatticeDecoder(pool) are linked against a TEST DOUBLE of the C ABI (tests/pool_double/fake_wfstdec.cc: a channel counts and
checksums the rows it is handed; two overlapping calls on one decoder abort -- the real library's calls are not re-entrant).
The driver (pool_tsan_main.cc) runs worker threads over ragged utterances in chunks through LogLikelihood pulls, with partial
results, and one thread that misuses its decoder: every utterance's result carries its own frames and rows, the misuse comes back as
an exception to that thread alone, the batcher batched, and the sanitizer has nothing to say (no race, no lock inversion, no
re-entrant call).  Round 6's hang -- advance requests never marked done -- fails here in seconds (the run is under a time-out)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "tests", "pool_double")


@pytest.fixture(scope="module")
def pool_tsan(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("tsan") / "pool_tsan")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-g", "-pthread", "-fsanitize=thread", os.path.join(D, "pool_tsan_main.cc"),
                           os.path.join(D, "fake_wfstdec.cc"), os.path.join(ROOT, "asr-decoder_amd", "host", "wfst-host.cc"), "-o", exe,
                           "-Wl,--unresolved-symbols=ignore-all"], stderr=subprocess.DEVNULL)
    return exe


@pytest.mark.parametrize("threads,utts", [(16, 96), (3, 20), (1, 5)])
def test_pool_batches_without_races(pool_tsan, threads, utts):
    p = subprocess.run([pool_tsan, str(threads), str(utts)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1"))
    assert "ThreadSanitizer" not in p.stderr, p.stderr[-3000:]
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr[-1500:])
    out = dict(zip(p.stdout.split()[0::2], p.stdout.split()[1::2]))
    assert out["bad"] == "0" and out["misuse_caught"] == "1"
    got, want = out["frames"].split("/")
    assert got == want
    if threads >= 8:
        assert int(out["advance_requests"]) >= 2 * int(out["advance_calls"])


def test_shared_device_mode_without_races(pool_tsan):
    """GpuLatticeDecoder::ShareDevice: decoder objects constructed the reference's way -- (graph, config), one per thread -- lease
    channels of shared device decoders (16 objects over two 8-channel ones): every utterance's own rows, the misuse to its own
    thread, nothing for the sanitizer."""
    p = subprocess.run([pool_tsan, "16", "96", "share"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1"))
    assert "ThreadSanitizer" not in p.stderr, p.stderr[-3000:]
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr[-1500:])
    assert "bad 0 misuse_caught 1" in p.stdout
