"""bench.py on the GPU box as the driver launches it: one rank of a torchrun-style environment, in a fresh child process
(VERDICT r2, next #4).  The first real N > 1 run happens on the driver's 8-GPU node; these tests remove what can be
removed of its risk on a one-GPU box: rank 0 of the RCCL path (device binding from LOCAL_RANK, barrier, the three
all-reduces), and -- as a marked REHEARSAL, both ranks on cuda:0 with the collectives over gloo, because RCCL refuses two
ranks on one GPU -- rank 1's row slice, timing and counter aggregation."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MOT_FORCE_DIST")}


def _one_json(stdout: str) -> dict:
    lines = [ln for ln in stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_bench_as_rank0_of_a_torchrun_environment():
    """RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 + MOT_FORCE_DIST=1: the process takes the N-rank code path over RCCL (init with
    device_id, barriers, all-reduce of the counters / the slowest time / the per-rank launch times)."""
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               MOT_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-extra", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _one_json(r.stdout)
    assert j["n_gpus"] == 1 and j["steps"] == 5 and j["unit"] == "tokens/s"
    roof = j["roofline"]
    assert len(roof["kernel_ms_per_rank"]) == 1 and abs(roof["kernel_ms_per_rank"][0] - roof["kernel_ms"]) < 1e-9
    assert 0.2 < roof["frac"] < 1.0 and roof["tokens_per_launch"] == 256 * 2048
    assert j["config"]["per_gpu_tokens"] == 256 * 2048 and j["config"]["global_tokens_per_step"] == 256 * 2048
    bs = j["byte_stats"]                       # the statistics pass ran once and went through the all-reduce
    assert bs["tokens"] == 256 * 2048 and bs["slots"] == 256 * 2048 * 16 and 0 < bs["pads_after"] < bs["pads_before"]
    assert roof["single_gpu_shard_reference"] is not None and roof["single_gpu_shard_reference"]["frac"] > 0.8


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_bench_two_ranks_rehearsal_on_one_device(scaling):
    """`bench.py --gpus 2 --backend gloo --one-device`: the self-launching parent (no torch import), two ranks that both run the
    real kernel on cuda:0, rank 1 owning rows [128, 256) of the one global batch (strong) -- the byte statistics summed over the
    ranks must equal the single-rank statistics of the whole batch, which pins rank 1's slice."""
    common = ["--steps", "4", "--warmup", "2", "--no-extra", "--no-cpu-baseline"]
    r1 = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "1"] + common, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-3000:]
    whole = _one_json(r1.stdout)["byte_stats"]
    r2 = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device", "--scaling", scaling] + common,
                        env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-3000:]
    j = _one_json(r2.stdout)
    assert j["n_gpus"] == 2 and j["rehearsal"] is True and j["scaling"] == scaling
    assert len(j["roofline"]["kernel_ms_per_rank"]) == 2 and all(t > 0 for t in j["roofline"]["kernel_ms_per_rank"])
    if scaling == "strong":
        assert j["config"]["per_gpu_tokens"] == 128 * 2048 and j["config"]["global_tokens_per_step"] == 256 * 2048
        for k in ("tokens", "slots", "pads_before", "pads_after"):
            assert j["byte_stats"][k] == whole[k], (k, j["byte_stats"], whole)
        assert j["roofline"]["single_gpu_shard_reference"]["kernel_us"] > 100      # the 262 144-token entry
    else:
        assert j["config"]["per_gpu_tokens"] == 256 * 2048 and j["byte_stats"]["tokens"] == 2 * 256 * 2048
