#!/usr/bin/env python3
"""Static extraction of the FleetRec embedding-kernel structure from the reference source.

TEST INFRASTRUCTURE / GENERATOR -- runs only in the build container (needs /root/reference).
It does NOT compile or execute the reference: the HLS headers (ap_int.h, hls_stream.h) are
absent from the image, so the reference kernels are unbuildable here.  Instead this script reads
the reference's text and symbolically walks the dataflow that decides the *record format*:

  * constants.hpp                       -> per-table DATA/PADDED/AXI sizes, row counts, start addrs
  * embedding_N_krnl(...) top level     -> which tables (and in which round order) each memory
                                           bank's load_single_embedding_K_tables<...> call serves
  * load_single_embedding_K_tables      -> rounds are emitted in template-argument order
  * group_* functions                   -> how 128-bit words are packed 4-at-a-time into 512-bit
                                           words (out.range(hi,lo) = tmpX)
  * gather_embeddings + gather_N_embedding_streams -> final order of 512-bit words per item
  * load_access_idx                     -> the 32 fixed indices

Output (committed): tests/golden/registry_{47,98,377}.json -- *data*, not reference source.
Each JSON lists the banks with their tables and, for every 128-bit word of the per-item record,
which (bank, k-th word of that bank's per-item stream) it carries.

Reference files walked (file:line of the main anchors are stored in the JSON as "anchors").
"""
import json
import os
import re
import sys

REF = os.environ.get("FLEETREC_REFERENCE", "/root/reference")
KDIR = os.path.join(REF, "FPGA/kernel/user_krnl")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden")


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"^[ \t]*#pragma[^\n]*", "", src, flags=re.M)
    return src


def parse_defines(path):
    defs = {}
    txt = strip_comments(open(path).read())
    for m in re.finditer(r"^[ \t]*#define[ \t]+(\w+)[ \t]+(.+?)[ \t]*$", txt, flags=re.M):
        name, val = m.group(1), m.group(2).strip()
        try:
            defs[name] = int(eval(val, {"__builtins__": {}}, dict(defs)))
        except Exception:
            pass
    return defs


def find_function(src, name):
    """Return (params_text, body_text, line_no) of `void name(...) {...}` (definition, not call)."""
    for m in re.finditer(r"\bvoid\s+" + re.escape(name) + r"\s*\(", src):
        i = m.end()
        depth = 1
        while depth:
            c = src[i]
            depth += (c == "(") - (c == ")")
            i += 1
        params = src[m.end():i - 1]
        j = i
        while src[j] in " \t\r\n":
            j += 1
        if src[j] != "{":
            continue
        k = j + 1
        depth = 1
        while depth:
            c = src[k]
            depth += (c == "{") - (c == "}")
            k += 1
        return params, src[j + 1:k - 1], src.count("\n", 0, m.start()) + 1
    raise KeyError(name)


def split_args(text):
    out, depth, cur = [], 0, ""
    for c in text:
        if c in "(<[":
            depth += 1
        elif c in ")>]":
            depth -= 1
        if c == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += c
    if cur.strip():
        out.append(cur.strip())
    return out


def parse_stream_params(params):
    """-> list of (kind, name, arraylen) with kind in {'axi','net','other'}."""
    res = []
    for p in split_args(params):
        m = re.match(r"hls::stream<\s*(\w+)\s*>\s*\(\s*&\s*(\w+)\s*\)\s*\[\s*(\d+)\s*\]", p)
        if m:
            res.append(("net" if m.group(1) == "network_t" else "axi", m.group(2), int(m.group(3))))
            continue
        m = re.match(r"hls::stream<\s*(\w+)\s*>\s*&\s*(\w+)", p)
        if m:
            res.append(("net" if m.group(1) == "network_t" else "axi", m.group(2), None))
            continue
        res.append(("other", p.split()[-1], None))
    return res


def walk_group_function(src, name):
    """Symbolically run one iteration of a group_* packer.

    Returns (input_param_names, output_param_names, emitted) where emitted is a list of
    (output_param_name, [slot0..slot3]) and slot = (input_param_position, k) meaning
    'the k-th 128-bit word read this iteration from that input stream'.
    """
    params, body, line = find_function(src, name)
    sp = parse_stream_params(params)
    ins = [n for k, n, _ in sp if k == "axi"]
    outs = [n for k, n, _ in sp if k == "net"]
    # body of the per-item for loop
    m = re.search(r"for\s*\([^)]*\)\s*\{", body)
    loop = body[m.end():body.rindex("}")]
    loop = re.sub(r"#pragma[^\n]*", "", loop)
    env, outw, counters, emitted = {}, {}, {n: 0 for n in ins}, []
    for st in [s.strip() for s in loop.split(";") if s.strip()]:
        m = re.match(r"axi_t\s+(\w+)\s*=\s*(\w+)\.read\(\)$", st)
        if m:
            s = m.group(2)
            env[m.group(1)] = (ins.index(s), counters[s])
            counters[s] += 1
            continue
        m = re.match(r"axi_t\s+(\w+)\s*=\s*(\w+)$", st)
        if m:
            env[m.group(1)] = env[m.group(2)]
            continue
        if re.match(r"network_t\s+\w+(\s*,\s*\w+)*$", st):
            continue
        m = re.match(r"(\w+)\.range\(\s*(\d+)\s*,\s*(\d+)\s*\)\s*=\s*(\w+)$", st)
        if m:
            hi, lo = int(m.group(2)), int(m.group(3))
            assert hi - lo == 127 and lo % 128 == 0, st
            outw.setdefault(m.group(1), [None] * 4)[lo // 128] = env[m.group(4)]
            continue
        m = re.match(r"(\w+)\.write\(\s*(\w+)\s*\)$", st)
        if m:
            w = outw[m.group(2)]
            assert None not in w, (name, st)
            emitted.append((m.group(1), list(w)))
            continue
        raise ValueError("unhandled statement in %s: %r" % (name, st))
    return ins, outs, emitted, counters, line


def walk_final_gather(src, name):
    """gather_N_embedding_streams: -> (array_param_names, [(array_name, j) ...] read order per item)."""
    params, body, line = find_function(src, name)
    sp = parse_stream_params(params)
    arrays = [(n, l) for k, n, l in sp if k == "net" and l is not None]
    m = re.search(r"for\s*\([^)]*\)\s*\{", body)
    inner = re.sub(r"#pragma[^\n]*", "", body[m.end():body.rindex("}")])
    order = []
    pos = 0
    # sequence of:  for (int j = 0; j < A; j++) { [for (int k = 0; k < B; k++) {] s_network.write(ARR[j].read()); }
    pat = re.compile(r"for\s*\(\s*int\s+j\s*=\s*0\s*;\s*j\s*<\s*(\d+)\s*;\s*j\+\+\s*\)\s*\{\s*"
                     r"(?:for\s*\(\s*int\s+k\s*=\s*0\s*;\s*k\s*<\s*(\d+)\s*;\s*k\+\+\s*\)\s*\{\s*)?"
                     r"s_network\.write\(\s*(\w+)\[j\]\.read\(\)\s*\)\s*;\s*\}\s*(\})?")
    while True:
        m = pat.search(inner, pos)
        if not m:
            break
        assert inner[pos:m.start()].strip() == "", inner[pos:m.start()]
        a, b, arr = int(m.group(1)), int(m.group(2) or 1), m.group(3)
        assert (m.group(2) is None) == (m.group(4) is None)
        for j in range(a):
            for _ in range(b):
                order.append((arr, j))
        pos = m.end()
    assert inner[pos:].strip() == "", inner[pos:]
    return arrays, order, line


def find_calls(body, fname_regex):
    """Yield (fname, template_args_text_or_None, args_text) for calls in body."""
    for m in re.finditer(r"\b(" + fname_regex + r")\s*(<)?", body):
        i = m.end()
        targs = None
        if m.group(2):
            depth = 1
            while depth:
                c = body[i]
                depth += (c == "<") - (c == ">")
                i += 1
            targs = body[m.end():i - 1]
        while body[i] in " \t\r\n":
            i += 1
        if body[i] != "(":
            continue
        j = i + 1
        depth = 1
        while depth:
            c = body[j]
            depth += (c == "(") - (c == ")")
            j += 1
        yield m.group(1), targs, body[i + 1:j - 1]


def extract(kname):
    hdir = os.path.join(KDIR, kname, "src", "hls")
    cpp_path = os.path.join(hdir, kname + ".cpp")
    raw = open(cpp_path).read()
    src = strip_comments(raw)
    defs = parse_defines(os.path.join(hdir, "constants.hpp"))
    anchors = {}

    # ---- load_single_embedding_K_tables: rounds emitted in template-argument order -------------
    for m in re.finditer(r"void\s+(load_single_embedding_(\d+)_tables)\s*\(", src):
        fn, n = m.group(1), int(m.group(2))
        _, body, line = find_function(src, fn)
        anchors[fn] = line
        seq = re.findall(r"base_addr_(\d+)\s*=\s*start_addr_(\d+)\s*\+\s*idx\s*\*\s*AXI_padded_size_(\d+)", body)
        assert [tuple(map(int, s)) for s in seq] == [(i, i, i) for i in range(n)], (fn, seq)
        wr = re.findall(r"for\s*\(\s*int\s+j\s*=\s*0\s*;\s*j\s*<\s*AXI_padded_size_(\d+)\s*;[^)]*\)\s*\{\s*"
                        r"s_embedding_buffer\.write\(\s*table_RAM\[\s*base_addr_(\d+)\s*\+\s*j\s*\]\s*\)", body)
        assert [tuple(map(int, s)) for s in wr] == [(i, i) for i in range(n)], (fn, wr)
        assert len(re.findall(r"s_idx_buffer\.read\(\)", body)) == 1, fn  # ONE idx per item per bank (F4)

    # ---- top-level kernel: bank -> tables ------------------------------------------------------
    _, top_body, line = find_function(src, kname)
    anchors[kname] = line
    banks = {}  # stream name -> bank dict
    for fn, targs, args in find_calls(top_body, r"load_single_embedding_\d+_tables"):
        a = split_args(args)
        idx_s, tab, emb_s = a[0], a[1], a[2]
        t = split_args(targs)
        assert len(t) % 2 == 0
        tables = []
        for addr_m, axi_m in zip(t[0::2], t[1::2]):
            ma = re.match(r"ADDR_AXI_(HBM|DDR|PLRAM)_(\d+)$", addr_m)
            mb = re.match(r"AXI_PADDED_SIZE_(HBM|DDR|PLRAM)_(\d+)$", axi_m)
            assert ma and mb and ma.groups() == mb.groups(), (addr_m, axi_m)
            cls, tid = ma.group(1), int(ma.group(2))
            tables.append({
                "name": "%s_%d" % (cls, tid), "class": cls, "id": tid,
                "addr_axi": defs[addr_m], "axi_words": defs[axi_m],
                "data_size": defs["DATA_SIZE_%s_%d" % (cls, tid)],
                "padded_size": defs["PADDED_SIZE_%s_%d" % (cls, tid)],
                "rows": defs["TABLE_SIZE_%s_%d" % (cls, tid)],
            })
        mb = re.match(r"table_(HBM|DDR|PLRAM)(\d+)$", tab)
        assert mb, tab
        assert idx_s == "s_idx_buffer_%s%s" % mb.groups() and emb_s == "s_embedding_buffer_%s%s" % mb.groups()
        bname = "%s%s" % mb.groups()
        size_macro = "%s_BANK%s_SIZE" % mb.groups()
        banks[emb_s] = {"name": bname, "class": mb.group(1), "bank": int(mb.group(2)),
                        "bank_axi_words": defs.get(size_macro), "tables": tables}
    # every idx stream is fed by load_access_idx (same sequence for every bank -- F4)
    n_idx = len(list(find_calls(top_body, r"load_access_idx")))
    assert n_idx == len(banks), (n_idx, len(banks))

    # ---- gather_embeddings call: positional mapping actual -> formal ---------------------------
    ge_params, ge_body, line = find_function(src, "gather_embeddings")
    anchors["gather_embeddings"] = line
    formals = [n for _, n, _ in parse_stream_params(ge_params)]
    calls = list(find_calls(top_body, r"gather_embeddings"))
    assert len(calls) == 1
    actuals = split_args(calls[0][2])
    assert len(actuals) == len(formals)
    formal_to_bank = {}
    for f, a in zip(formals, actuals):
        if a in banks:
            formal_to_bank[f] = a

    # ---- walk gather_embeddings: group calls fill level_A FIFOs --------------------------------
    fifos = {}  # (array, j) -> list of words, each word = 4 x (bank_stream, k)
    final_name = None
    final_args = None
    group_cache = {}
    for fn, _, args in find_calls(ge_body, r"group_\w+|gather_\d+_embedding_streams"):
        a = split_args(args)
        if fn.startswith("gather_"):
            final_name, final_args = fn, a
            continue
        if fn not in group_cache:
            group_cache[fn] = walk_group_function(src, fn)
            anchors[fn] = group_cache[fn][4]
        ins, outs, emitted, counters, _ = group_cache[fn]
        assert len(a) == len(ins) + len(outs) + 1 and a[-1] == "batch_num", (fn, a)
        act_in = a[:len(ins)]
        act_out = a[len(ins):len(ins) + len(outs)]
        # words consumed per input stream must equal that bank's per-item stream length
        for pos, formal in enumerate(act_in):
            bank = banks[formal_to_bank[formal]]
            want = sum(t["axi_words"] for t in bank["tables"])
            assert counters[ins[pos]] == want, (fn, formal, counters[ins[pos]], want)
        for oname, slots in emitted:
            tgt = act_out[outs.index(oname)]
            m = re.match(r"(\w+)\[(\d+)\]$", tgt)
            key = (m.group(1), int(m.group(2)))
            word = [(formal_to_bank[act_in[p]], k) for p, k in slots]
            fifos.setdefault(key, []).append(word)

    arrays, order, line = walk_final_gather(src, final_name)
    anchors[final_name] = line
    # positional: formal array name -> actual array name
    arr_map = {f: a for (f, _), a in zip(arrays, final_args)}
    heads = {k: 0 for k in fifos}
    record = []
    for arr, j in order:
        key = (arr_map[arr], j)
        w = fifos[key][heads[key]]
        heads[key] += 1
        record.extend(w)
    for k in fifos:
        assert heads[k] == len(fifos[k]), ("unconsumed FIFO words", k)

    bank_list = sorted(banks.values(), key=lambda b: ({"HBM": 0, "DDR": 1, "PLRAM": 2}[b["class"]], b["bank"]))
    bank_pos = {b["name"]: i for i, b in enumerate(bank_list)}
    rec = [[bank_pos[banks[s]["name"]], k] for s, k in record]

    # ---- load_access_idx: the 32 fixed indices -------------------------------------------------
    _, body, line = find_function(src, "load_access_idx")
    anchors["load_access_idx"] = line
    m = re.search(r"idx_random\s*\[\s*\]\s*=\s*\{([^}]*)\}", body)
    idx_random = [int(x) for x in m.group(1).split(",")]

    assert len(rec) == defs["INPUT_SIZE_AXI_512"] * 4
    return {
        "kernel": kname,
        "source": "FPGA/kernel/user_krnl/%s/src/hls/%s.cpp + constants.hpp" % (kname, kname),
        "derivation": "static extraction (symbolic walk of the reference text); not produced by executing the reference",
        "anchors": anchors,
        "input_size": defs["INPUT_SIZE"],
        "record_words_512": defs["INPUT_SIZE_AXI_512"],
        "fpga_batch_size": defs["BATCH_SIZE"],
        "fc": ([defs["INPUT_SIZE"], defs["HIDDEN_SIZE1"], defs["HIDDEN_SIZE2"], defs["HIDDEN_SIZE3"], defs["OUTPUT_SIZE"]]
               if "HIDDEN_SIZE1" in defs else None),  # 377: FC dims live in the 3-node GPU constant.h
        "idx_random": idx_random,
        "banks": bank_list,
        "record": rec,
    }


def extract_gpu():
    """GPU-side compile-time configuration (constant.h of the two final servers)."""
    g = {}
    d1 = parse_defines(os.path.join(REF, "GPU/final_network_cublasLt_1_node_no_FIFO_scatter/constant.h"))
    g["one_node"] = {
        "source": "GPU/final_network_cublasLt_1_node_no_FIFO_scatter/constant.h:21-42",
        "fc": [d1["INPUT_FEATURE_LEN"], d1["HIDDEN_SIZE1"], d1["HIDDEN_SIZE2"], d1["HIDDEN_SIZE3"], d1["OUTPUT_FEATURE_LEN"]],
        "batch_size": d1["BATCH_SIZE"], "total_batch_num": d1["TOTAL_BATCH_NUM"],
        "port": d1["PORT"], "thread_num": d1["THREAD_NUM"],
    }
    d3 = parse_defines(os.path.join(REF, "GPU/final_network_cublasLt_3_nodes_no_FIFO_scatter/constant.h"))
    g["three_nodes"] = {
        "source": "GPU/final_network_cublasLt_3_nodes_no_FIFO_scatter/constant.h:25-60",
        "fc": [d3["INPUT_FEATURE_LEN_RECEIVER"], d3["HIDDEN_SIZE1"], d3["HIDDEN_SIZE2"], d3["HIDDEN_SIZE3"], d3["OUTPUT_FEATURE_LEN"]],
        "len_fpga_sender": d3["INPUT_FEATURE_LEN_FPGA_SENDER"], "len_cpu_sender": d3["INPUT_FEATURE_LEN_CPU_SENDER"],
        "batch_size": d3["BATCH_SIZE"], "total_batch_num": d3["TOTAL_BATCH_NUM"], "thread_num": d3["THREAD_NUM"],
        "ports": [d3["PORT_CPU_SENDER_0"], d3["PORT_FPGA_SENDER_0"], d3["PORT_FPGA_SENDER_1"]],
        # cuda_server.c:515,541,566 (3-node): the receive buffer is three concatenated blocks
        "block_order": ["CPU0", "FPGA0", "FPGA1"],
    }
    # README known answers (GPU/final_network_cublasLt_1_node_no_FIFO_scatter/README.md:7-11)
    g["known_answers"] = [{"k": 512, "fc": [512, 1024, 512, 256, 1], "score": 68719476736},
                          {"k": 1024, "fc": [1024, 1024, 512, 256, 1], "score": 137438953472}]
    return g


def main():
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "registry_gpu.json"), "w") as f:
        json.dump(extract_gpu(), f, indent=1)
        f.write("\n")
    for n in (47, 98, 377):
        reg = extract("embedding_%d_krnl" % n)
        path = os.path.join(OUT, "registry_%d.json" % n)
        with open(path, "w") as f:
            json.dump(reg, f, indent=None, separators=(",", ":"))
            f.write("\n")
        ntab = sum(len(b["tables"]) for b in reg["banks"])
        print("%s: %d banks, %d tables, %d record words(128b), idx_random=%d -> %s" % (
            reg["kernel"], len(reg["banks"]), ntab, len(reg["record"]), len(reg["idx_random"]), os.path.relpath(path)))


if __name__ == "__main__":
    sys.exit(main())
