"""End-to-end shape of the reference restored: the TCP request server (THREAD_NUM connections on
PORT+i, fixed-size blocks, mutex-guarded batch counter) fed by the synthetic sender, with indices on the wire."""
import os
import re
import subprocess
import time

import pytest
from conftest import free_port_block

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "gpu-fpga-recommendation-system_amd", "host")


@pytest.mark.parametrize("model,val,extra", [("A", 352.0 * 2 ** 27, []), ("C", 3968.0 * 2 ** 28, ["--row-cap", "200"]),
                                             ("A", 352.0 * 2 ** 27, ["--stream"]),
                                             ("A", 352.0 * 2 ** 27, ["--stream", "--reply"]),   # scores back over the socket, sender window 64
                                             ("B", 880.0 * 2 ** 27, ["--per-bank", "--row-cap", "200"]),          # the kernel's own index contract on the wire
                                             ("C", 3968.0 * 2 ** 28, ["--row-cap", "200", "--shards", "1"]),      # sharded engine + RCCL (one-rank communicator)
                                             ("C", 3968.0 * 2 ** 28, ["--row-cap", "200", "--shards", "3", "--one-device"])])   # three shard contexts on one GPU: the staged exchange
def test_server_and_sender_known_answer(fr, gpu, model, val, extra):
    """Reference data end to end: even/odd tables, the 32 fixed indices, all-ones weights -> the first five scores of the
    last batch are 0 0 K*H1*H2*H3 K*H1*H2*H3 0 (indices 3, 99, 38, 72, 29), as the reference prints them (cuda_server.c:499-502)."""
    if not os.path.exists(os.path.join(HOST, "fleetrec_server")):
        subprocess.check_call(["make", "-s", "-C", HOST])
    threads, total = 4, 64
    port = free_port_block(threads)
    srv = subprocess.Popen([os.path.join(HOST, "fleetrec_server"), "--model", model, "--batch", "128", "--threads", str(threads),
                            "--port", str(port), "--total", str(total), "--tables", "evenodd", "--weights", "ones"] + extra,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    time.sleep(0.5)
    snd = subprocess.Popen([os.path.join(HOST, "fleetrec_sender"), "--model", model, "--batch", "128", "--threads", str(threads),
                            "--port", str(port), "--indices", "reference"] + [e_ for e_ in extra if e_ not in ("--stream", "--shards", "1", "3", "--one-device")]
                           + (["--window", "64"] if "--reply" in extra else []),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    try:
        out, _ = srv.communicate(timeout=300)
        sout, _ = snd.communicate(timeout=60)
    finally:
        for p in (srv, snd):
            if p.poll() is None:
                p.kill()
    out = out.decode()
    assert srv.returncode == 0, out
    assert "processed %d batches" % total in out, out
    assert re.search(r"first connection -> last scores: [0-9.]+ s = [0-9.]+ M inferences/s over TCP", out), out
    rows = re.findall(r"thread \d+ scores:((?: [-0-9.e+]+)+)", out)
    assert 1 <= len(rows) <= threads   # a thread that connected after the last batch was taken prints no scores
    assert len(rows) + out.count("took no batch") == threads
    for r in rows:
        v = [float(x) for x in r.split()]
        assert v == [0.0, 0.0, val, val, 0.0], (v, out)
    if "--stream" not in extra and "--shards" not in extra:
        assert "Average time from batch received to enqueued" in out
    if "--shards" in extra:
        assert ("table-sharded over 3 shard contexts on GPU" if "--one-device" in extra else "table-sharded over 1 GPUs") in out
    assert "blocks sent" in sout.decode()
    if "--reply" in extra:      # every request the server took was answered, and the sender timed the round trips
        m_ = re.search(r"latency request sent -> scores received  n=(\d+) avg ([0-9.]+) us", sout.decode())
        assert m_ and int(m_.group(1)) >= 16, sout.decode()


def test_latency_measurement_mode(fr, gpu):
    """the reference's latency experiment (measure_network_cuda_cp_latency_single_node/cuda_server.c):
    a rate-limited sender, per-batch 'received' -> 'enqueued on the device' -> 'scores on the host' times and their summary."""
    if not os.path.exists(os.path.join(HOST, "fleetrec_server")):
        subprocess.check_call(["make", "-s", "-C", HOST])
    threads, total = 2, 80
    port = free_port_block(threads)
    srv = subprocess.Popen([os.path.join(HOST, "fleetrec_server"), "--model", "A", "--batch", "256", "--threads", str(threads), "--port", str(port),
                            "--total", str(total), "--tables", "hash", "--weights", "uniform", "--latency"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    time.sleep(0.5)
    snd = subprocess.Popen([os.path.join(HOST, "fleetrec_sender"), "--model", "A", "--batch", "256", "--threads", str(threads), "--port", str(port),
                            "--interval-us", "1000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    try:
        out, _ = srv.communicate(timeout=300)
        snd.communicate(timeout=60)
    finally:
        for p in (srv, snd):
            if p.poll() is None:
                p.kill()
    out = out.decode()
    assert srv.returncode == 0, out
    m1 = re.search(r"latency batch received -> enqueued\s+n=(\d+) avg ([0-9.]+) us", out)
    m2 = re.search(r"latency batch received -> scores on host\s+n=(\d+) avg ([0-9.]+) us\s+p50 ([0-9.]+)", out)
    assert m1 and m2, out
    assert total - 8 * threads <= int(m2.group(1)) <= total - 8   # up to the first 8 batches of each thread are dropped as warm-up
    assert float(m1.group(2)) <= float(m2.group(2))          # enqueue returns before the scores exist
    assert 0.0 < float(m2.group(3)) < 20000.0, out           # a batch of 256 answers within 20 ms even on a cold box
    assert "i = 0 recv->enqueued" in out


def test_stream_reply_answers_everything_before_a_clean_eof(fr, gpu):
    """a sender that is done half-closes its connections (--window W > 1: shutdown(SHUT_WR)) and still expects the replies of
    every request it sent.  With fewer requests than the server's --total (--max-blocks below the per-connection share), every
    connection ends on EOF: the server must treat EOF-before-a-request as a clean end of stream -- answer all accepted requests, take
    no counter slot for a request that never comes, exit 0."""
    if not os.path.exists(os.path.join(HOST, "fleetrec_server")):
        subprocess.check_call(["make", "-s", "-C", HOST])
    threads, total, per_conn = 4, 64, 10
    port = free_port_block(threads)
    srv = subprocess.Popen([os.path.join(HOST, "fleetrec_server"), "--model", "A", "--batch", "128", "--threads", str(threads), "--port", str(port),
                            "--total", str(total), "--tables", "evenodd", "--weights", "ones", "--stream", "--reply"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    time.sleep(0.5)
    snd = subprocess.Popen([os.path.join(HOST, "fleetrec_sender"), "--model", "A", "--batch", "128", "--threads", str(threads), "--port", str(port),
                            "--indices", "reference", "--reply", "--window", "4", "--max-blocks", str(per_conn)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    try:
        out, _ = srv.communicate(timeout=300)
        sout, _ = snd.communicate(timeout=60)
    finally:
        for p in (srv, snd):
            if p.poll() is None:
                p.kill()
    out, sout = out.decode(), sout.decode()
    assert srv.returncode == 0, out
    assert "processed %d batches" % (threads * per_conn) in out, out
    assert "sender: %d blocks sent" % (threads * per_conn) in sout, sout
    m_ = re.search(r"latency request sent -> scores received  n=(\d+) avg", sout)
    assert m_ and int(m_.group(1)) == threads * per_conn, sout     # every request was answered
    for r in re.findall(r"thread \d+ scores:((?: [-0-9.e+]+)+)", out):
        assert [float(x) for x in r.split()] == [0.0, 0.0, 352.0 * 2 ** 27, 352.0 * 2 ** 27, 0.0], out
