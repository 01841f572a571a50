"""A fixed handful of the randomised cases of tests/fuzz_vs_oracle.py inside the suite: random model family / guide / shape / batch
source / clipping threshold / noise scale / step count, HIP trajectory vs the oracle's (tolerances in that file).  Long sweeps are run
by hand (`python tests/fuzz_vs_oracle.py first count`); profiles/ keeps their records."""
import pytest

from . import fuzz_vs_oracle as F

pytestmark = pytest.mark.gpu


# (0 .. 39, and the seeds of the sweep's first 440 that found something: 23 -- Gaussian mean with rows too wide for the register-tiled
#  kernel; 119, 229, 335, 349 -- the loss of a FIRST empty batch read the updated parameters (k_flush in place); 259, 310, 326 -- the loss of
#  an empty batch behind a NaN state on the column-chunked / two-kernel path; 275, 337 -- losses that are small differences of large sums)
@pytest.mark.parametrize("seed", list(range(40)) + [119, 229, 259, 275, 310, 326, 335, 337, 349])
def test_random_case_vs_oracle(gpu, O, seed):
    c = F.run_case(F.draw_case(seed), O)
    assert c["ok"], c


# (18, 37, 54: an empty batch behind a NaN state -- the reference's masked sum is NaN * 0 = NaN; 18 is also the case whose NaN
#  concentration made the Marsaglia-Tsang loop of the Gamma sampler spin: it is bounded now, on the device and in the oracle)
@pytest.mark.parametrize("seed", list(range(24)) + [37, 54, 73])
def test_random_mixture_model_case_vs_oracle(gpu, O, seed):
    c = F.run_gmm_case(F.draw_gmm_case(seed), O)
    assert c["ok"], c


@pytest.mark.parametrize("seed", list(range(20)))
def test_random_vae_case_vs_oracle(gpu, O, seed):
    c = F.run_vae_case(F.draw_vae_case(seed), O)
    assert c["ok"], c


@pytest.mark.parametrize("seed", list(range(30)))
def test_random_rng_and_sampler_arguments_vs_oracle(gpu, O, seed):
    c = F.run_rng_case(F.draw_rng_case(seed), O)
    assert c["ok"], c


# (the five-stage composition with SGD, then evaluate.  84, 107: masked rows of the column-chunked materialising kernel behind a NaN state;
#  176, 297: a scale below 6e-8 -- softplus(u) for u < -16.6 was 0 on the device: 1 + exp(u) rounds to 1; 95, 210: a row with a NaN entry is
#  NaN throughout after clipping, jnp.maximum propagates NaN; 236: exp-parametrised scales of e^27 -- __expf carries |u| ulp)
@pytest.mark.parametrize("seed", list(range(30)) + [84, 95, 107, 153, 176, 210, 236, 297])   # (a third of them with ADADP)
def test_random_stage_composition_case_vs_oracle(gpu, O, seed):
    c = F.run_staged_case(F.draw_staged_case(seed), O)
    assert c["ok"], c


# (the data-parallel split with the ranks emulated one after another: 1 .. 8 row shards, Feistel or Poisson batches, the one-launch engine or
#  the two-kernel one, vs the ORACLE's single-device trajectory.  20, 66, 68, 72, 90: rows too wide for the register-tiled kernel through
#  FusedHipEngine -- d3p_dpvi_logreg_fused_step launched the one-launch kernel with the column-chunked geometry and returned NaN; the
#  engine now takes the two-kernel steps there and the C entry refuses.  57, 78: the loss of a first, suppressed Poisson batch from the
#  flush launch, which reads the arrays it publishes to)
@pytest.mark.parametrize("seed", list(range(24)) + [57, 66, 68, 72, 78, 90, 103, 210])   # (103, 210: the native loop REFUSES rows it has no step form for)
def test_random_row_sharded_case_vs_oracle(gpu, O, seed):
    c = F.run_shards_case(F.draw_shards_case(seed), O)
    assert c["ok"], c


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_batchifier_case_vs_oracle(gpu, O, seed):
    c = F.run_batches_case(F.draw_batches_case(seed), O)
    assert c["ok"], c


@pytest.mark.parametrize("seed", list(range(20)))
def test_random_position_sharded_case_vs_the_whole_batch_update(gpu, O, seed):
    c = F.run_posshards_case(F.draw_posshards_case(seed), O)
    assert c["ok"], c


# (batches of 5000 .. 100 000 examples -- several examples per wave, more workgroups than CUs --, random prior and likelihood scales)
@pytest.mark.parametrize("seed", list(range(12)))
def test_random_large_batch_case_vs_oracle(gpu, O, seed):
    c = F.run_case(F.draw_big_case(seed), O)
    assert c["ok"], c


# (run_steps over the batchifiers that have no native loop: sampling with replacement, the epoch split)
@pytest.mark.parametrize("seed", list(range(16)))
def test_random_stepwise_batchifier_case_vs_oracle(gpu, O, seed):
    c = F.run_case(F.draw_stepwise_case(seed), O)
    assert c["ok"], c
