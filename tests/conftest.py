import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "autotuned: a whole-step test that WANTS the timing-based kernel picks (HIP against HIP, bit for bit)")


# Collection order of the GPU suite.  The driver runs `pytest -m gpu -x`: whatever fails first hides everything behind it, so
# the suite runs from the sharpest evidence to the bluntest -- (0) single kernels against torch / float64 on the CPU (conv
# variants, training ops, split-K hand-off, optimizer kernels), (1) the one-rank RCCL choreography, (2) whole models against
# the reference-made fixtures and the oracle, (3) whole training steps against fixtures, (4) multi-step / statistical
# behaviour (fits a batch, trajectories, loops).  Inside a tier the file / definition order is kept.
_FILE_TIER = {
    "test_ops_gpu.py": 0, "test_conv_probe_gpu.py": 0, "test_conv_fuzz_gpu.py": 0, "test_train_ops_gpu.py": 0,
    "test_cedt_gpu.py": 0, "test_episode_io_gpu.py": 0, "test_regularisers_gpu.py": 0, "test_properties_gpu.py": 0,
    "test_bf16_variant_gpu.py": 0,
    "test_stage1_gpu.py": 2, "test_models_gpu.py": 2, "test_panet_gpu.py": 2, "test_eval_protocol_gpu.py": 2,
    "test_autograd_bridge_gpu.py": 2, "test_dataset_gpu.py": 2, "test_grad_frozen_gpu.py": 3, "test_train_gpu.py": 3,
}
_TEST_TIER = {
    "test_optimizer_step_matches_torch_sgd": 0, "test_fused_adam_step_matches_torch_adam": 0,
    "test_overlapped_gradient_buckets_on_one_rank_match_the_plain_step": 1, "test_bench_collectives_run_on_rccl_with_one_rank": 1,
    "test_two_steps_reduce_loss_and_dropblock_runs": 4, "test_training_fits_a_fixed_batch": 4,
    "test_training_loop_epochs_eval_and_checkpoints": 4, "test_five_step_trajectory_matches_the_reference": 4,
    "test_train_command_writes_a_fresh_run_and_test_command_loads_it": 4,
}


def pytest_collection_modifyitems(config, items):
    def tier(item):
        name = getattr(item, "originalname", None) or item.name.split("[")[0]
        return _TEST_TIER.get(name, _FILE_TIER.get(os.path.basename(str(item.fspath)), 2))
    order = {id(it): i for i, it in enumerate(items)}
    items.sort(key=lambda it: (tier(it), order[id(it)]))
    if os.environ.get("PEMP_TEST_REVERSE"):          # order-dependence check: the same suite back to front, one process
        items.reverse()


@pytest.fixture(scope="session")
def hip_lib():
    """The C-ABI library; built on demand (hipcc cross-compiles without a GPU)."""
    from pemp_amd import build, _lib
    build.build()
    return _lib.load()


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


#: modules whose tests compare whole training steps with reference-made fixtures at a tolerance
_PINNED_MODULES = {"test_train_gpu.py", "test_grad_frozen_gpu.py", "test_panet_gpu.py", "test_autograd_bridge_gpu.py", "test_cedt_gpu.py",
                   "test_regularisers_gpu.py"}


@pytest.fixture(autouse=True)
def pinned_picks(request, monkeypatch):
    """Whole-step parity tests run ONE fixed kernel variant per layer (pemp_amd.ops.AUTOTUNE off: the 64 x 64 tile, no split-K,
    the library's 768-block weight-gradient split) from EMPTY pick caches.  The autotuners choose by timing, so their picks
    differ from box to box, and the split-K / weight-gradient splits regroup float32 sums: with timing-based picks a step
    that is green on one box says nothing about the next one (a fresh box landed on the other side of a
    bound five builder runs had met).  Every kernel reduces in a fixed order, so with the picks pinned the same test computes
    the same bits on every MI355X.  The admissible picks are swept, against the one-step bound, by
    test_train_gpu.py::test_train_step_under_every_admissible_pick; tests marked ``autotuned`` (HIP against HIP, bit for bit)
    keep the timing-based picks, which is what they are about."""
    if os.path.basename(str(request.node.fspath)) not in _PINNED_MODULES or request.node.get_closest_marker("autotuned"):
        yield None
        return
    from pemp_amd import ops
    saved = (dict(ops._TILE_CACHE), dict(ops.WGRAD_PICKS))
    ops._TILE_CACHE.clear()
    ops.WGRAD_PICKS.clear()
    monkeypatch.setattr(ops, "AUTOTUNE", False)
    monkeypatch.setattr(ops, "PICK_HOOK", None)
    yield ops
    ops._TILE_CACHE.clear()
    ops.WGRAD_PICKS.clear()
    ops._TILE_CACHE.update(saved[0])
    ops.WGRAD_PICKS.update(saved[1])


@pytest.fixture(autouse=True)
def fixed_torch_seed(request):
    """Every GPU test starts from the same torch seed.  The training engines seed their Philox stream (DropBlock / Dropout2d
    draws) from torch.initial_seed() at construction: without this a test's draws -- and the outcome of a stochastic test --
    depended on which tests had run before it in the process (round 6: a test that passed in the suite's order failed alone)."""
    if request.node.get_closest_marker("gpu"):
        import torch
        torch.manual_seed(20260105)
    yield


@pytest.fixture
def exact_eval_variants(monkeypatch):
    """Evaluation convs restricted to the variants that are bit-identical to each other (no split-K for small row counts:
    pemp_amd.ops.EVAL_SPLITK): what the "one episode per step equals the batched step bit for bit" tests state."""
    from pemp_amd import ops
    monkeypatch.setattr(ops, "EVAL_SPLITK", False)
