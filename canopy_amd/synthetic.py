"""Synthetic inputs for the Monte-Carlo path (host utility; the DAG is a small model, generated with
numpy and handed to the ABI as plain arrays -- canopy_dag_view)."""
import numpy as np

from . import _capi as capi


def layered_dag(seed, n_events, n_layers, width, fan_in, atleast_every=0):
    """Layered AND/OR gate DAG: layer 0 reads basic events, layer l reads layer l-1 (and a few events),
    OR and AND layers alternate (layer 0 = OR), every `atleast_every`-th gate is a K-of-N vote; the top
    gate is an OR over the last layer. Arguments are distinct per gate; about one in eight argument of
    an AND gate is complemented (a complemented rare event under an OR would pin the tree to TRUE)."""
    rng = np.random.default_rng(seed)
    gate_type, gate_min, arg_ptr, args = [], [], [0], []
    prev = list(range(n_events))
    g = 0
    for layer in range(n_layers):
        cur = []
        for _ in range(width):
            pool = prev if layer == 0 else prev + list(rng.choice(n_events, size=min(2, n_events), replace=False))
            k = min(fan_in, len(pool))
            a = rng.choice(pool, size=k, replace=False)
            t = capi.GATE_OR if layer % 2 == 0 else capi.GATE_AND
            neg = (rng.random(k) < 0.125) & (t == capi.GATE_AND)
            m = 0
            if atleast_every and g % atleast_every == atleast_every - 1 and k >= 2:
                t, m = capi.GATE_ATLEAST, int(rng.integers(2, k + 1))
                neg[:] = False
            gate_type.append(t)
            gate_min.append(m)
            args.extend(int(~x) if n else int(x) for x, n in zip(a, neg))
            arg_ptr.append(len(args))
            cur.append(n_events + g)
            g += 1
        prev = cur
    gate_type.append(capi.GATE_OR)
    gate_min.append(0)
    args.extend(prev)
    arg_ptr.append(len(args))
    top = n_events + g
    return dict(n_events=n_events, gate_type=np.array(gate_type, dtype=np.int32), gate_min=np.array(gate_min, dtype=np.int32),
                arg_ptr=np.array(arg_ptr, dtype=np.int32), args=np.array(args, dtype=np.int32), top=top)
