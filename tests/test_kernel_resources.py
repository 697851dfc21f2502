"""Register budgets of the hot kernels, read from the code objects inside libttenv.so (no GPU needed).

The loop's speed hangs on a few occupancy facts that no numerics test sees (DESIGN.md sections 4.1, 4.2): the policy kernel must not
spill (one wave per SIMD, all 512 registers: a spilled build ran at 0.144 ms per step instead of 0.085), the weight-gradient
kernels must fit THREE workgroups on a CU (<= 168 registers: beside the policy's 171 workgroups 85 CUs are free for their 200-210;
the actor's variant sat at 192 for three rounds and its last 30 workgroups started 5 us late), the row kernels two waves per SIMD
(eight waves per workgroup), the env step four."""
import pytest

from ddpg_trucktrailer_amd import kernel_resources as kr


@pytest.fixture(scope="module")
def ks():
    out = kr.kernels()
    assert len(out) >= 40, "the library's gfx950 code objects were not found"
    return out


def the(ks, *parts):
    found = kr.find(ks, *parts)
    assert found, f"no kernel named like {parts}"
    return found


def test_no_kernel_spills_vector_registers(ks):
    bad = {n: v["vgpr_spills"] for n, v in ks.items() if v["vgpr_spills"]}
    assert not bad, bad


def test_policy_kernel_takes_one_wave_per_simd_without_scratch_traffic(ks):
    for n, v in the(ks, "k_mlp_split").items():
        assert v["vgpr"] <= 512 and v["vgpr_spills"] == 0, (n, v)
        assert v["scratch"] <= 16, (n, v)          # (a few spilled SGPRs' worth of lanes, no vector spill)


def test_weight_gradient_kernels_fit_three_workgroups_per_cu(ks):
    for n, v in the(ks, "k_bwd_weights").items():
        assert v["vgpr"] <= 168 and kr.waves_per_simd(v["vgpr"]) >= 3, (n, v)
        assert v["scratch"] == 0 and 3 * v["lds"] <= 160 * 1024, (n, v)


def test_row_kernels_fit_their_eight_waves_on_a_cu(ks):
    for part in ("k_fwd_multi", "k_fwd_small", "k_bwd_rows_pair", "k_actor_tail"):
        for n, v in the(ks, part).items():
            assert kr.waves_per_simd(v["vgpr"]) >= 2 and v["scratch"] == 0 and v["lds"] <= 160 * 1024, (n, v)


def test_env_step_of_the_loop_keeps_four_waves_per_simd(ks):
    # k_step<PER_ENV = false, INFO = false, AUTO_RESET, RANDOM_POLICY>: the variants bench.py and the DDPG loop launch
    for n, v in the(ks, "6k_stepILb0ELb0E").items():
        assert kr.waves_per_simd(v["vgpr"]) >= 4 and v["scratch"] == 0, (n, v)
