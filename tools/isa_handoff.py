"""Build-time check of the fence-free level-to-level hand-off of k_track_quad<.., LEVELS = true> (variant 7).

That hand-off is correct by what gfx950 does, not by the HIP memory model: relaxed agent-scope accesses ordered by the
wave's own instruction order (csrc/pagk_quad_kernel.h).  Nothing in the language keeps a
compiler from changing what the argument rests on, so the shape is checked in the assembly the library is built from
(`hipcc -save-temps=obj`), between the `; pagk-handoff:` comment markers the kernel emits:

  state begin .. state end        every global store carries sc1 (agent scope: written through to the coherence point)
  publish begin .. publish end    an `s_waitcnt vmcnt(0)` comes before the first memory instruction, i.e. every state store
                                  has been acknowledged before the slot is drawn; the slot is a returning
                                  `global_atomic_add`; a second `s_waitcnt vmcnt(0)` sits between it and the entry store;
                                  the entry store carries sc1
  take-over begin .. take-over end  at least four loads, all `global_load_dword .. sc1`, and no fence
                                  (`buffer_inv` / `buffer_wbl2`) inside any marked region: the design measured them out
... and the polling loop in front of the take-over: the load next to `s_sleep 0x7f` carries sc1.

Every instantiation with LEVELS = true must contain each region at least once.  Usage:
    python tools/isa_handoff.py <assembly.s>        (exit code 0 = the shape holds; __graft_entry__.build_hip calls check())
"""
import re
import sys

# k_track_quad<NCH, LEAN, LEVELS = true, BATCH>
KERNEL = re.compile(r"^(_ZN4pagk12k_track_quadILi\d+ELb[01]ELb1ELb[01]EEEvNS_9TrackArgsE):")
MEM = re.compile(r"^\s*(global_|buffer_|flat_|scratch_)")


def kernels(lines):
    """{name: [instruction lines]} of the LEVELS instantiations"""
    out, cur = {}, None
    for ln in lines:
        m = KERNEL.match(ln)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is not None:
            if ln.startswith(".Lfunc_end"):
                cur = None
            else:
                cur.append(ln.rstrip())
    return out


def regions(body, name):
    """instruction lists between `; pagk-handoff: <name> begin` and `... end`"""
    out, cur = [], None
    for ln in body:
        if f"; pagk-handoff: {name} begin" in ln:
            cur = []
        elif f"; pagk-handoff: {name} end" in ln:
            if cur is not None:
                out.append(cur)
            cur = None
        elif cur is not None:
            t = ln.strip()
            if t and not t.startswith((";", ".", "//")) and not t.endswith(":"):
                cur.append(t)
    return out


def check(asm_text):
    """list of violations (empty = the shape holds)"""
    bad = []
    ks = kernels(asm_text.splitlines())
    if not ks:
        return ["no k_track_quad<.., LEVELS = true> instantiation in the assembly"]
    for k, body in ks.items():
        where = k[len("_ZN4pagk12"):len("_ZN4pagk12") + 32]

        def no_fence(region, what):
            for ins in region:
                if ins.startswith(("buffer_inv", "buffer_wbl2")):
                    bad.append(f"{where}: a cache fence ({ins.split()[0]}) inside {what}")

        st = regions(body, "state")
        if not st:
            bad.append(f"{where}: no `state` region")
        for r in st:
            stores = [i for i in r if i.startswith("global_store") or i.startswith("flat_store")]
            if len(stores) < 4:
                bad.append(f"{where}: {len(stores)} state stores (4 expected)")
            for i in stores:
                if " sc1" not in i:
                    bad.append(f"{where}: state store without agent scope: {i}")
            no_fence(r, "the state stores")
        pub = regions(body, "publish")
        if not pub:
            bad.append(f"{where}: no `publish` region")
        for r in pub:
            mem = [(n, i) for n, i in enumerate(r) if MEM.match(i) or i.startswith("s_waitcnt")]
            seq = [i for _, i in mem if not (i.startswith("s_waitcnt") and "vmcnt(0)" not in i)]
            kinds = ["W" if i.startswith("s_waitcnt") else ("A" if "atomic_add" in i else ("S" if "_store_" in i else "?")) for i in seq]
            shape = "".join(kinds)
            # wait, atomic (returning), wait, store -- further waits in between are harmless
            if not re.fullmatch(r"W+AW+S", shape):
                bad.append(f"{where}: publish sequence is {shape or '(empty)'} (W+ A W+ S expected: {seq})")
                continue
            atomic = next(i for i in seq if "atomic_add" in i)
            if len(atomic.split(",")) < 4:
                bad.append(f"{where}: the slot atomic does not return a value: {atomic}")
            store = next(i for i in seq if "_store_" in i)
            if " sc1" not in store:
                bad.append(f"{where}: the publishing store is not agent scope: {store}")
            no_fence(r, "the publication")
        tk = regions(body, "take-over")
        if not tk:
            bad.append(f"{where}: no `take-over` region")
        for r in tk:
            loads = [i for i in r if "_load_" in i and not i.startswith("scratch")]
            if len(loads) < 4:
                bad.append(f"{where}: {len(loads)} state loads (4 expected)")
            for i in loads:
                if " sc1" not in i or not i.startswith("global_load_dword"):
                    bad.append(f"{where}: state load without agent scope: {i}")
            no_fence(r, "the take-over")
        # the poll: the load(s) within a few instructions of the long sleep
        ins = [l.strip() for l in body]
        sleeps = [n for n, i in enumerate(ins) if i.startswith("s_sleep 0x7f")]
        if not sleeps:
            bad.append(f"{where}: no polling loop (s_sleep 0x7f)")
        for n in sleeps:
            near = [i for i in ins[max(0, n - 12):n + 12] if i.startswith("global_load_dword")]
            if not near:
                bad.append(f"{where}: no load next to the polling loop's sleep")
            for i in near:
                if " sc1" not in i:
                    bad.append(f"{where}: polling load without agent scope: {i}")
    return bad


if __name__ == "__main__":
    problems = check(open(sys.argv[1]).read())
    for p in problems:
        print("isa_handoff:", p)
    print("isa_handoff: %s" % ("FAILED" if problems else "ok"))
    sys.exit(1 if problems else 0)
