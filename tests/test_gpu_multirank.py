"""-m gpu: the N > 1 path of the ENGINE backend on one GPU — two ranks (processes) share the card and exchange their finished
reference pictures over gloo (the rehearsal mode of bench.py: the transfers are staged through the host, everything else — group
buffers, one message per peer and wave, the exchange stream and its events, reference-half bookkeeping, the picture check against
the CPU checker at the end — is the production code path that RCCL runs on N GPUs)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(900)
@pytest.mark.parametrize("extra", [[], ["--exchange", "allgather"], ["--gop", "ldp"], ["--gop", "intra"], ["--scaling", "strong"], ["--host-threads", "1"]],
                         ids=["ra_readers", "ra_allgather", "ldp", "intra", "strong", "one_host_thread"])
def test_two_ranks_on_one_gpu(extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "480p_main8",
           "--chains", "4", "--streams", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=850, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["check"]["ok"], out["check"]
    assert out["scaling"] == ("strong" if "strong" in extra else "weak")
    ex = out["config"]["exchange_per_rank"]
    assert len(ex) == 2
    if "intra" in extra:
        assert all(e["MB_sent_per_step"] == 0 for e in ex)
    else:
        assert all(e["MB_sent_per_step"] > 0 and e["MB_received_per_step"] > 0 for e in ex)
        if "allgather" in extra:
            assert all(e["collectives_per_step"] > 0 and e["messages_per_step"] == 0 for e in ex)
        else:
            # one message per peer, wave and stream: 2 streams x 4 waves x (1 send + 1 receive) at most
            assert all(0 < e["messages_per_step"] <= 2 * 4 * 2 for e in ex)
    assert out["host_threads"] == (1 if "--host-threads" in extra else 2)       # a host thread per stream, exchanges ordered by P.Turnstile
    chains = 2 if "strong" in extra else 4
    assert out["config"]["chains_in_flight_per_gpu"] == chains
