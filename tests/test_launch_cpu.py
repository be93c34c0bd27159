"""`bench.py --gpus N` starts N ranks itself (tcdiff_amd/launch.py).  This drives the SAME launcher function bench.py
uses with world_size 2 over gloo and the stub rank body (process group, shard ranges, all-reduce, JSON relay; no GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launcher_starts_two_ranks_and_relays_one_json_line(capfd):
    bench = _bench()
    rc = bench.launch_ranks(["--gpus", "2", "--stub", "--batch", "5"], 2, timeout=300)
    out = capfd.readouterr().out
    assert rc == 0
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out                       # rank 1's stdout is not relayed
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2
    assert res["clip_ranges"] == [[0, 5], [5, 10]]
    assert res["local_rank"] == 0
    assert abs(res["max_time"] - 0.002) < 1e-9        # MAX over ranks


def test_spawned_ranks_get_distinct_local_ranks():
    sys.path.insert(0, ROOT)
    from tcdiff_amd import launch
    code = "import os,sys; sys.stdout.write(os.environ['RANK']+os.environ['LOCAL_RANK']+os.environ['WORLD_SIZE']); " \
           "sys.stderr.write('r'+os.environ['LOCAL_RANK'])"
    rc, out0, errs = launch.spawn_ranks(["-c", code], 3)
    assert rc == 0 and out0 == "003"
    assert [e for e in errs] == ["r0", "r1", "r2"]


def test_failed_rank_fails_the_launch():
    sys.path.insert(0, ROOT)
    from tcdiff_amd import launch
    code = "import os,sys; sys.exit(3 if os.environ['RANK']=='1' else 0)"
    rc, _, _ = launch.spawn_ranks(["-c", code], 2)
    assert rc == 3


def test_mismatched_world_size_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--stub"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "WORLD_SIZE=2" in p.stderr


def test_single_rank_default_is_not_a_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["n_gpus"] == 1 and res["ranks_seen"] == 1 and res["clip_ranges"] == [[0, 16]]


def test_eight_rank_launch_as_the_driver_runs_it(capfd):
    """world 8 (the SCALE run's size): 8 processes here need ~8 x 0.6 GB of torch import, fine for this container"""
    bench = _bench()
    rc = bench.launch_ranks(["--gpus", "8", "--stub", "--batch", "16"], 8, timeout=600)
    out = capfd.readouterr().out
    assert rc == 0
    res = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 8 and res["ranks_seen"] == 8
    assert res["clip_ranges"] == [[16 * r, 16 * r + 16] for r in range(8)]
    assert abs(res["max_time"] - 0.008) < 1e-9


def test_a_chatty_or_dead_peer_cannot_hang_the_launch():
    """ranks used to be drained one at a time through pipes.  Rank 1 writes 1 MB to stderr (a pipe holds ~64 KB)
    while rank 0 waits for it; then a peer that dies must take the survivors down instead of leaving them waiting."""
    sys.path.insert(0, ROOT)
    from tcdiff_amd import launch
    import time
    chatty = ("import os,sys,time\n"
              "r=os.environ['RANK']\n"
              "p='/tmp/tcdiff_launch_test_'+os.environ['MASTER_PORT']\n"
              "if r=='1':\n"
              "    sys.stderr.write('x'*(1<<20)); sys.stderr.flush(); open(p,'w').write('done')\n"
              "else:\n"
              "    t=time.time()\n"
              "    while not os.path.exists(p) and time.time()-t<60: time.sleep(0.05)\n"
              "    sys.stdout.write('ok' if os.path.exists(p) else 'stuck')\n")
    rc, out0, errs = launch.spawn_ranks(["-c", chatty], 2, timeout=120)
    assert rc == 0 and out0 == "ok" and len(errs[1]) == 2000
    dead = "import os,sys,time\nif os.environ['RANK']=='1': sys.exit(5)\ntime.sleep(600)\n"
    t0 = time.time()
    rc, _, _ = launch.spawn_ranks(["-c", dead], 2, timeout=300)
    assert rc == 5 and time.time() - t0 < 30
