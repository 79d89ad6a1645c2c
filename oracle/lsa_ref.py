"""TEST INFRASTRUCTURE ONLY (see oracle/eda_oracle.h): CPU restatement of the assignment problem
the reference's HungarianMatcher delegates to scipy (models/losses.py:319-329,
`linear_sum_assignment(c[i])` on the (Q, T_b) cost block of scene i).

scipy is a third-party dependency of the reference (requirements: scipy; this image pins 1.15.3)
and not part of /root/reference.  Its published algorithm (scipy/optimize/_lsap.py ->
rectangular_lsap.cpp: Crouse, "On implementing 2D rectangular assignment algorithms", 2016) is a
shortest-augmenting-path method with dual potentials.  This file restates that method in plain
numpy (fp64 potentials) for the orientation the matcher uses -- more queries than targets, every
target gets a distinct query -- and tests pin it against scipy itself on random and adversarial
matrices (tests/test_losses.py).  Parity status: pinned by scipy 1.15.3 (present in this image);
the optimum is unique unless costs tie.
"""
import numpy as np


def assign_targets(cost):
    """cost: (Q, T) array, T <= Q.  Returns q (T,) int64: q[t] = query assigned to target t,
    minimising sum_t cost[q[t], t]."""
    cost = np.asarray(cost, dtype=np.float64)
    Q, T = cost.shape
    assert T <= Q
    INF = float("inf")
    u = np.zeros(T + 1)
    v = np.zeros(Q + 1)
    p = np.zeros(Q + 1, dtype=np.int64)      # p[j] = target row matched to query column j (1-based)
    way = np.zeros(Q + 1, dtype=np.int64)
    for i in range(1, T + 1):
        p[0] = i
        j0 = 0
        minv = np.full(Q + 1, INF)
        used = np.zeros(Q + 1, dtype=bool)
        while True:
            used[j0] = True
            i0 = p[j0]
            free = ~used
            free[0] = False
            cur = cost[:, i0 - 1] - u[i0] - v[1:]
            better = free[1:] & (cur < minv[1:])
            idx = np.nonzero(better)[0] + 1
            minv[idx] = cur[idx - 1]
            way[idx] = j0
            cand = np.where(free, minv, INF)
            j1 = int(np.argmin(cand))             # lowest column on ties
            delta = cand[j1]
            u[p[used]] += delta
            v[used] -= delta
            minv[free] -= delta
            j0 = j1
            if p[j0] == 0:
                break
        while True:
            j1 = way[j0]
            p[j0] = p[j1]
            j0 = j1
            if j0 == 0:
                break
    q = np.full(T, -1, dtype=np.int64)
    for j in range(1, Q + 1):
        if p[j]:
            q[p[j] - 1] = j - 1
    return q
