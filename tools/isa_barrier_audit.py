"""Every s_barrier of every kernel in libsehip.so must have an `s_waitcnt ... lgkmcnt(0)` in front of it (only ALU work in between).

Why: a work-group barrier orders LDS traffic only for stores that have LEFT the wave's LDS queue.  The compiler's waitcnt pass
left one barrier of the LDS bitonic sort (loop header, reached over the back edge) without that wait and se_topk_rows returned
unsorted rows whenever a second stream kept the CU's LDS pipe busy.  `wg_barrier()` (se_common.h) now
issues the wait itself; this audit reads the shipped code objects, so a kernel that goes back to a bare __syncthreads() -- or
a compiler that moves the wait -- is caught on the CPU, without a GPU.

    python tools/isa_barrier_audit.py [path/to/libsehip.so]      # exit 1 and the offending kernels when a barrier is bare
    python tools/isa_barrier_audit.py --dataflow <lib or code object> [kernel regex]   # path-sensitive check for foreign code
"""
import os, re, struct, subprocess, sys, tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    """The gfx950 ELF images of the uncompressed clang offload bundles embedded in a host shared object."""
    blob = open(path, "rb").read()
    out = []
    for m in re.finditer(MAGIC, blob):
        base = m.start()
        (n,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, pos)
            triple = blob[pos + 24:pos + 24 + tlen].decode()
            pos += 24 + tlen
            if "amdgcn" in triple and size:
                out.append((triple, blob[base + off:base + off + size]))
    return out


def kernel_code_bytes(path):
    """{demangled-or-mangled kernel name: code bytes} of every function symbol in the gfx950 code objects of `path` (round 6: the
    per-row loop of the ranking kernels must stay inside the 64 KB instruction cache, profiles/r06_b_rank_icache.txt)."""
    sizes = {}
    for triple, image in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(image); f.flush()
            text = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-sW", f.name], check=True, capture_output=True, text=True).stdout
        for ln in text.split("\n"):
            fld = ln.split()
            if len(fld) >= 8 and fld[3] == "FUNC" and fld[2].isdigit():
                sizes[fld[7]] = max(sizes.get(fld[7], 0), int(fld[2]))
    return sizes


def audit(path):
    """-> (kernels, barriers, bare) with bare = [(kernel, address, reason)].

    A barrier passes when, walking back from it, an `s_waitcnt` with lgkmcnt(0) is met before any LDS instruction, any
    branch and any branch target (the scheduler may slide ALU work between the wait and the barrier; nothing else)."""
    kernels, barriers, bare = set(), 0, []
    images = [image for triple, image in code_objects(path)]
    if not images and open(path, "rb").read(4) == b"\x7fELF":
        images = [None]                                # `path` is a bare gfx950 code object (e.g. unbundled from another library)
    for image in images:
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            if image is not None:
                f.write(image); f.flush()
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name if image is not None else path], check=True, capture_output=True, text=True).stdout
        funcs, cur = [], None                          # [(name, start address, [(address, instruction)])]
        for ln in text.split("\n"):
            m = re.match(r"^([0-9a-f]+) <([^>]+)>:", ln)
            if m:
                cur = (m.group(2), int(m.group(1), 16), [])
                funcs.append(cur)
                continue
            m = re.match(r"^\s+(\S.*?)\s+// ([0-9A-F]+): ", ln)
            if m and cur is not None:
                cur[2].append((int(m.group(2), 16), m.group(1), ln))
        for name, start, body in funcs:
            targets = set()
            for addr, ins, ln in body:
                if ins.startswith("s_branch") or ins.startswith("s_cbranch"):
                    m = re.search(r"<[^>]+\+0x([0-9a-f]+)>\s*$", ln)
                    targets.add(start + int(m.group(1), 16) if m else -1)
            for i, (addr, ins, ln) in enumerate(body):
                if not ins.startswith("s_barrier"):
                    continue
                barriers += 1
                kernels.add(name)
                reason, j = "start of kernel", i
                while j > 0:
                    if body[j][0] in targets:
                        reason = "branch target at %x before a wait" % body[j][0]; break
                    j -= 1
                    p = body[j][1]
                    if p.startswith("s_waitcnt") and "lgkmcnt(0)" in p:
                        reason = None; break
                    if p.startswith("ds_") or p.startswith("s_branch") or p.startswith("s_cbranch") or p.startswith("s_barrier"):
                        reason = "reaches `%s` first" % p; break
                if reason:
                    bare.append((name, "%x" % addr, reason))
    return kernels, barriers, bare


def disassemble(path):
    """[(kernel, start address, [(address, instruction, raw line)])] of every function of every gfx950 code object in `path`."""
    images = [image for triple, image in code_objects(path)]
    if not images and open(path, "rb").read(4) == b"\x7fELF":
        images = [None]
    funcs = []
    for image in images:
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            if image is not None:
                f.write(image); f.flush()
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name if image is not None else path], check=True, capture_output=True, text=True).stdout
        cur = None
        for ln in text.split("\n"):
            m = re.match(r"^([0-9a-f]+) <([^>]+)>:", ln)
            if m:
                cur = (m.group(2), int(m.group(1), 16), [])
                funcs.append(cur)
                continue
            m = re.match(r"^\s+(\S.*?)\s+// ([0-9A-F]+): ", ln)
            if m and cur is not None:
                cur[2].append((int(m.group(2), 16), m.group(1), ln))
    return funcs


LDS_STORE = re.compile(r"^ds_(write|store|add|sub|rsub|inc|dec|min|max|and|or|xor|mskor|cmpst|cmpswap|wrxchg|wrap|append|consume|pk_add|condxchg|storexchg)")


def hazards(path, only=None):
    """Control-flow version of the question, for code this repository does not own (no wg_barrier() there): is there a PATH from an
    LDS store / atomic to an s_barrier with no `s_waitcnt lgkmcnt(0)` on it?  Forward data flow over the basic blocks of each kernel
    (state = "a store of this wave may still be queued"), iterated to the fixed point.  -> [(kernel, barrier address, store address)]."""
    out = []
    for name, start, body in disassemble(path):
        if only and not re.search(only, name):
            continue
        n = len(body)
        if not n:
            continue
        index = {a: i for i, (a, _, _) in enumerate(body)}
        succ = [[] for _ in range(n)]
        for i, (addr, ins, ln) in enumerate(body):
            if ins.startswith("s_endpgm") or ins.startswith("s_setpc") or ins.startswith("s_swappc"):
                continue
            if ins.startswith("s_branch") or ins.startswith("s_cbranch"):
                m = re.search(r"<[^>]+\+0x([0-9a-f]+)>\s*$", ln)
                t = index.get(start + int(m.group(1), 16)) if m else None
                if t is not None:
                    succ[i].append(t)
                if ins.startswith("s_branch"):
                    continue
            if i + 1 < n:
                succ[i].append(i + 1)
        pending = [None] * n                      # address of a store that may be queued when instruction i STARTS, or None
        work = [0]
        seen = [False] * n
        while work:
            i = work.pop()
            st = pending[i]
            ins = body[i][1]
            if ins.startswith("s_waitcnt") and "lgkmcnt(0)" in ins:
                st = None
            elif LDS_STORE.match(ins):
                st = body[i][0]
            for j in succ[i]:
                if not seen[j] or (pending[j] is None and st is not None):
                    seen[j] = True
                    if st is not None or pending[j] is None:
                        pending[j] = st if pending[j] is None else pending[j]
                    work.append(j)
        for i, (addr, ins, ln) in enumerate(body):
            if ins.startswith("s_barrier") and pending[i] is not None:
                out.append((name, "%x" % addr, "%x" % pending[i]))
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(here, "semantic-embeddings_amd", "sehip", "libsehip.so")
    if "--dataflow" in sys.argv:
        hz = hazards(lib, args[1] if len(args) > 1 else None)
        print("%s: %d barriers reachable from an LDS store with no lgkmcnt(0) wait on the path" % (os.path.basename(lib), len(hz)))
        for kern, addr, st in hz[:60]:
            print("  %s: barrier at %s, store at %s" % (kern[:110], addr, st))
        sys.exit(1 if hz else 0)
    k, b, bare = audit(lib)
    print("%s: %d barriers in %d kernels, %d without the LDS wait" % (os.path.basename(lib), b, len(k), len(bare)))
    for kern, addr, why in bare[:40]:
        print("  bare barrier in %s at %s: %s" % (kern[:90], addr, why))
    sys.exit(1 if bare else 0)
