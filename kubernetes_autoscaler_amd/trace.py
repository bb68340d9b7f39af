"""Call traces of the ENCODER half of the C ABI: every casim_enc_* call a host makes while it walks its pods and node
templates, written as one text line per call.  tools/casim_native (plain C++, no Python, no torch) replays such a trace
through libcasim.so and times encode -> finalize -> upload -> kernels -> fetch natively: the figures a cgo shim would see
(the Python mirror spends most of a call in ctypes and object walking).

Line format: name TAB arg TAB arg ...   strings escaped (\\t \\n \\\\), NULL = "~"; arrays = count followed by the items.
Lines starting with '@' are directives for the engine phase (written by the helpers at the bottom)."""
import contextlib
import ctypes as C

from . import _abi

_SKIP = {"casim_enc_destroy", "casim_enc_tables", "casim_enc_domain_rules", "casim_enc_port_block", "casim_enc_dict_sizes",
         "casim_enc_lane_count", "casim_enc_lane_name"}   # (read-only)
# integer array arguments: function -> {arg index: length spec}; 'R' = encoder lanes (MAX_RES slots are passed, R are read),
# ('arg', i) = value of argument i, ('mul', i, 'R') = argument i times R, None = output / unused
_ARRAYS = {
    "casim_enc_add_group": {2: "R"},
    "casim_enc_add_pod_spec": {2: "R"},
    "casim_enc_group_set_pegs": {2: ("arg", 3)},
    "casim_enc_add_resource_pegs": {3: ("mul", 2, "R"), 4: ("arg", 2), 5: None},
}


def _esc(b):
    if b is None:
        return "~"
    s = b.decode("utf-8", "replace") if isinstance(b, (bytes, bytearray)) else str(b)
    return s.replace("\\", "\\\\").replace("\t", "\\t").replace("\n", "\\n")


class CallTrace:
    def __init__(self):
        self.lines = []
        self.n_res = 2

    def directive(self, name, *tokens):
        self.lines.append("\t".join(["@" + name] + [str(t) for t in tokens]))

    def save(self, path):
        with open(path, "w") as f:
            f.write("\n".join(self.lines) + "\n")

    # ---- recording ---------------------------------------------------------------------------------
    def record(self, name, args):
        if name in _SKIP or not name.startswith("casim_enc_"):
            return
        if name == "casim_enc_create":
            o = args[0]._obj if hasattr(args[0], "_obj") else args[0]
            self.n_res = int(o.n_res)
            self.lines.append("\t".join([name, str(o.n_res), str(o.enable_taint_comparison_ops), str(o.explicit_self_exclusion)]))
            return
        if name == "casim_enc_add_running_pods":
            # (e, n_pods, group[n], ns[n], req[n][R], label_off[n + 1], label_key[L], label_val[L], strings[S], S)
            n, S = int(args[1]), int(args[9])
            L = int(args[5][n]) if n > 0 else 0

            def ints(a, count):
                return [str(count)] + [str(int(a[k])) for k in range(count)]
            toks = [name, str(n)] + ints(args[2], n) + ints(args[3], n) + ints(args[4], n * self.n_res) + ints(args[5], n + 1) + ints(args[6], L) + ints(args[7], L)
            toks += [str(S)] + [_esc(args[8][k]) for k in range(S)]
            self.lines.append("\t".join(toks))
            return
        if name == "casim_enc_add_pods":
            self.lines.append("\t".join([name] + self._pod_columns(args[1]._obj if hasattr(args[1], "_obj") else args[1].contents)))
            return
        argtypes = _abi.PROTOTYPES[name][1]
        arrays = _ARRAYS.get(name, {})
        toks = [name]
        for i, (t, a) in enumerate(zip(argtypes, args)):
            if i == 0 and t is C.c_void_p:
                continue   # the encoder handle
            if t in (C.c_int32, C.c_int64):
                toks.append(str(int(a)))
            elif t is C.c_double:
                toks.append(repr(float(a)))
            elif t is _abi.cstr:
                toks.append(_esc(a))
            elif t is _abi.cstrp:
                n = next(int(args[j]) for j in range(i + 1, len(args)) if argtypes[j] is C.c_int32)
                toks.append(str(n))
                for k in range(n):
                    toks.append(_esc(a[k]))
            elif t in (_abi.i64p, _abi.i32p):
                spec = arrays.get(i, "missing")
                if spec is None or a is None:
                    toks.append("0")
                    continue
                if spec == "missing":
                    raise ValueError(f"{name}: argument {i} needs an array length rule")
                if spec == "R":
                    n = self.n_res
                elif spec[0] == "arg":
                    n = int(args[spec[1]])
                else:
                    n = int(args[spec[1]]) * self.n_res
                toks.append(str(n))
                toks.extend(str(int(a[k])) for k in range(n))
            else:
                raise ValueError(f"{name}: cannot serialise argument {i} of type {t}")
        self.lines.append("\t".join(toks))


    def _pod_columns(self, pc):
        """casim_pod_columns as tokens: n_pods, the string table, then every column as (count, items...) — 0 items = a NULL column."""
        n = int(pc.n_pods)

        def arr(ptr, count, fmt=lambda x: str(int(x))):
            if not ptr or count <= 0:
                return ["0"]
            return [str(count)] + [fmt(ptr[k]) for k in range(count)]
        toks = [str(n), str(int(pc.n_strings))] + [_esc(pc.strings[k]) for k in range(int(pc.n_strings))]
        toks += arr(pc.ns, n) + arr(pc.req, n * self.n_res) + arr(pc.fastpath_req, 2 * n, lambda x: repr(float(x))) + arr(pc.peg_count, n)
        for off, cols in ((pc.label_off, (pc.label_key, pc.label_val)), (pc.tol_off, (pc.tol_key, pc.tol_op, pc.tol_value, pc.tol_effect)),
                          (pc.sel_off, (pc.sel_key, pc.sel_val))):
            total = int(off[n]) if off and n > 0 else 0
            toks += arr(off, n + 1 if off else 0)
            for c in cols:
                toks += arr(c, total)
        return toks


class _RecordingLib:
    def __init__(self, lib, trace):
        self._lib, self._trace = lib, trace

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("casim_enc_"):
            return fn

        def call(*args):
            self._trace.record(name, args)
            return fn(*args)
        return call


@contextlib.contextmanager
def recording():
    """with recording() as trace: ... every Encoder built inside is traced (one encoder per trace)."""
    from . import encoder
    trace = CallTrace()
    real = encoder.lib
    encoder.lib = _RecordingLib(real, trace)
    try:
        yield trace
    finally:
        encoder.lib = real


# ---- engine-phase directives ------------------------------------------------------------------------
def _arr(a):
    a = [] if a is None else list(a)
    return [len(a)] + [int(x) for x in a]


def add_estimate(trace, kinds=(0,), iters=20, fastpath=False):
    """casim_estimate_batch_timed on the finalized tables, one simulation holding every group."""
    trace.directive("estimate", int(fastpath), iters, *_arr(kinds))


def add_try_schedule(trace, pod_class, hint_node=None, node_acceptable=None, break_on_failure=False, last_index=0, similar_key=None,
                     use_rules=True, iters=5):
    trace.directive("try_schedule", iters, int(break_on_failure), int(last_index), int(use_rules), *(_arr(pod_class) + _arr(hint_node) +
                    _arr(node_acceptable) + _arr(similar_key)))


def add_removals(trace, cand_node, pod_offsets, pod_class, hint_node=None, destination=None, persist=True, max_removable=0,
                 last_index=0, use_rules=True, iters=5):
    trace.directive("removals", iters, int(persist), int(max_removable), int(last_index), int(use_rules),
                    *(_arr(cand_node) + _arr(pod_offsets) + _arr(pod_class) + _arr(hint_node) + _arr(destination)))
