"""The hand-out protocol of the one-level-per-wave kernel (pagk_quad_kernel.h, LEVELS), restated as a small
discrete-event model and run under adversarial conditions: any number of resident slots (down to ONE), waves made
resident in any order and on any XCD, random run times.  What the kernel's comment claims must hold in every run:
the launch drains (no wave waits for ever), every (quad, level) item is run exactly once, a quad's levels run in
order, and a consumer never waits for a producer that has not started.

This is a model of the protocol, not of the arithmetic (the GPU tests check the bits); it needs no device."""
import random

import pytest


def simulate(n_quads, levels, slots, seed, max_events=10**7):
    rng = random.Random(seed)
    cnt = [(n_quads - x + 7) // 8 for x in range(8)]               # quads of sequence x (q = x mod 8)
    cmax = (n_quads + 7) // 8
    tickets = [0] * 8                                              # a.queue[32 x]
    finished = [[0] * levels for _ in range(8)]                    # a.queue[32 x + 1 + step]
    ready = [[[0] * cmax for _ in range(8)] for _ in range(levels)]  # a.lv_ready[(step * 8 + x) * cmax + slot]
    waves_left = n_quads * levels                                  # the grid: one wave per item
    resident = []                                                  # dicts: state of a wave that holds a slot
    ran = {}                                                       # (quad, level step) -> order of execution
    clock = 0
    order = 0
    started_tickets = [set() for _ in range(8)]

    def take_ticket(xcc):
        for k in range(8):
            x = (xcc + k) & 7
            t = tickets[x]
            tickets[x] += 1
            if t < cnt[x] * levels:
                started_tickets[x].add(t)
                return x, t // cnt[x], t - (t // cnt[x]) * cnt[x]
        return None

    events = 0
    while waves_left > 0 or resident:
        events += 1
        assert events < max_events, "the launch does not drain"
        # the dispatcher: fill free slots with new waves, each on a random XCD (the order of waves is immaterial: a
        # wave has no identity before it takes its ticket)
        while waves_left > 0 and len(resident) < slots:
            waves_left -= 1
            item = take_ticket(rng.randrange(8))
            assert item is not None, "a wave found no item although the grid has as many waves as items"
            x, step, j = item
            resident.append({"x": x, "step": step, "j": j, "quad": 8 * j + x if step == 0 else None, "end": None})
        # waves that wait for their ready-list entry look again
        progressed = False
        for w in resident:
            if w["quad"] is None:
                e = ready[w["step"] - 1][w["x"]][w["j"]]
                if e:
                    w["quad"] = e - 1
                    progressed = True
                else:
                    # the claim behind the liveness argument: every ticket of the step above has been taken
                    lo, hi = (w["step"] - 1) * cnt[w["x"]], w["step"] * cnt[w["x"]]
                    assert all(t in started_tickets[w["x"]] for t in range(lo, hi)), "a consumer waits for a producer that has not started"
            if w["quad"] is not None and w["end"] is None:
                w["end"] = clock + rng.randint(1, 40)
                progressed = True
        running = [w for w in resident if w["end"] is not None]
        assert running or progressed, "every resident wave waits: deadlock"
        if not running:
            continue
        clock = min(w["end"] for w in running)
        for w in [w for w in running if w["end"] <= clock]:
            key = (w["quad"], w["step"])
            assert key not in ran, f"item {key} ran twice"
            assert 0 <= w["quad"] < n_quads and w["quad"] % 8 == w["x"]
            if w["step"] > 0:
                assert (w["quad"], w["step"] - 1) in ran, "a level ran before the level above had finished"
            ran[key] = order
            order += 1
            if w["step"] < levels - 1:     # lv_publish
                slot = finished[w["x"]][w["step"]]
                finished[w["x"]][w["step"]] += 1
                ready[w["step"]][w["x"]][slot] = w["quad"] + 1
            resident.remove(w)
    assert len(ran) == n_quads * levels
    return ran


@pytest.mark.parametrize("n_quads,levels,slots", [(1, 3, 1), (5, 3, 1), (5, 3, 2), (8, 2, 3), (9, 4, 4), (37, 3, 5),
                                                  (64, 3, 64), (100, 3, 16), (100, 5, 7), (257, 3, 300), (500, 3, 41)])
def test_the_launch_drains_and_every_item_runs_once_in_level_order(n_quads, levels, slots):
    for seed in range(12):
        simulate(n_quads, levels, slots, seed)


def test_a_single_slot_runs_the_items_in_ticket_order():
    # one resident wave at a time: nobody can ever wait, because everything with a lower ticket has finished
    ran = simulate(23, 3, 1, 7)
    for (q, s), o in ran.items():
        if s > 0:
            assert ran[(q, s - 1)] < o
