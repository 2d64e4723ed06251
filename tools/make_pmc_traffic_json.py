"""profiles/r0N_pmc_traffic.json from the per-counter PMC summaries in profiles/ (tools/publish_profiles_round.sh runs this):
traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB * 1024 — FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies
128-B read requests at 64 B).   python tools/make_pmc_traffic_json.py [r03 | r04 | r05]"""
import json
import os
import re
import sys

P = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
RND = sys.argv[1] if len(sys.argv) > 1 else "r04"
B_STEP = 1723            # algorithmic bytes per env-step of k_step2 with f32 observations
# everything k_act_step<OBS_U8> moves per env-step: r03 = ig + hg given separately; r04 = one gate tensor + the masked hidden
# rows for the next step's GEMM; r05 = r04 without the activated gates (bench.py: policy_state_included)
B_ACT = {"r03": 16086, "r04": 13014}.get(RND, 8918)
_ACT_KERNEL = {"r03": "t2d::k_act_step<OBS_U8> (ig + hg given separately: the 4096-env timed region of round 3)",
              "r04": "t2d::k_act_step<OBS_U8> (one gate tensor, bias added in the kernel, masked hidden rows written: the timed "
                     "region from 768 envs up since round 4)",
              "r05": "t2d::k_act_step<OBS_U8> (round 4's form without the activated-gates store: the rollout keeps the gate GEMM's "
                     "output for the learner instead; tools/act_step_bench.py ACT_BENCH_MODE=pre)"}


ACT_KERNEL = _ACT_KERNEL.get(RND, _ACT_KERNEL["r05"])


def mean(name):
    return float(re.search(r"mean=([0-9.]+)", open(os.path.join(P, name)).read()).group(1))


def entry(n, prefix, algo):
    f, w = mean("%s_%s_pmc_FETCH_SIZE_%d.txt" % (RND, prefix, n)), mean("%s_%s_pmc_WRITE_SIZE_%d.txt" % (RND, prefix, n))
    return f, w, int(round((2 * f + w) * 1024)), algo * n


f, w, t, a = entry(4096, "env_only", B_STEP)
out = {"formula": "(2 * FETCH_SIZE + WRITE_SIZE) KiB * 1024 — FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B "
                  "read requests at 64 B)",
       "source": "profiles/%s_env_only_pmc_{FETCH,WRITE}_SIZE_<N>.txt, profiles/%s_act_step_pmc_{FETCH,WRITE}_SIZE_4096.txt "
                 "(separate rocprofv3 --pmc passes, tools/collect_profiles_round.sh %s)" % (RND, RND, RND),
       "n_envs": 4096, "kernel": "t2d::k_step2 (f32 observations)", "fetch_size_kib_per_launch": f,
       "write_size_kib_per_launch": w, "traffic_bytes_per_launch": t, "algorithmic_bytes_per_launch": a}
f, w, t, a = entry(4096, "act_step", B_ACT)
out["act_step"] = {"kernel": ACT_KERNEL, "one_gate_tensor": RND != "r03",
                   "fetch_size_kib_per_launch": f, "write_size_kib_per_launch": w, "traffic_bytes_per_launch": t,
                   "bytes_moved_by_the_kernel_per_launch": a, "survey_8d_bytes_per_launch": 709 * 4096}
out["other_sizes"] = {}
for n in (262144, 1048576):
    try:
        f, w, t, a = entry(n, "env_only", B_STEP)
        out["other_sizes"][str(n)] = {"fetch_kib": f, "write_kib": w, "traffic_bytes": t, "algorithmic_bytes": a}
    except (IOError, OSError, AttributeError):
        pass
json.dump(out, open(os.path.join(P, "%s_pmc_traffic.json" % RND), "w"), indent=1)
print(json.dumps(out)[:300])
