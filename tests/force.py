"""BT_FORCE (batrack_amd/csrc/ba_plan.hpp): the one environment switch of the kernel / solver / planner choices, read once per
process — so the tests that force a choice run a child process.  Helpers to read it and to add tokens to it."""
import os


def tokens(env=None):
    """{key: value} of the tokens in force ("kernel=k_edge2,prec=f32" -> {"kernel": "k_edge2", "prec": "f32"})."""
    e = os.environ if env is None else env
    out = {}
    for t in e.get("BT_FORCE", "").split(","):
        if t:
            k, _, v = t.partition("=")
            out[k] = v
    return out


def env_with(*toks, env=None):
    """A copy of the environment with these tokens added to BT_FORCE (a token replaces one of the same key)."""
    e = dict(os.environ if env is None else env)
    cur = tokens(e)
    for t in toks:
        k, _, v = t.partition("=")
        cur[k] = v
    e["BT_FORCE"] = ",".join(f"{k}={v}" for k, v in cur.items())
    return e


def f32_edges(env=None):
    """The per-edge maths of this environment's plans is float32 / mixed on the small fixtures too (round-2 gates apply)."""
    t = tokens(env)
    return t.get("prec") == "f32" or t.get("kernel") in ("k_stream", "k_edge2")
