// Built-in model descriptions (generated data) + validation / scaling of user-supplied ones.
// Counterpart of the reference's compile-time configuration: the generated per-kernel constants.hpp
// (FPGA/kernel/user_krnl/embedding_{47,98,377}_krnl/src/hls/constants.hpp) and the GPU servers'
// constant.h (GPU/final_network_cublasLt_{1_node,3_nodes}_no_FIFO_scatter/constant.h) become one
// run-time fr_model_desc.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

#include "fr_internal.h"
#include "registry_data.inc"

extern "C" const fr_model_desc *fr_model_builtin(int which) {
    switch (which) {
        case FR_MODEL_A: return &fr_model_a_desc;
        case FR_MODEL_B: return &fr_model_b_desc;
        case FR_MODEL_C: return &fr_model_c_desc;
        default: fr_set_error("fr_model_builtin: unknown model %d", which); return nullptr;
    }
}

int fr_model_validate(const fr_model_desc *m) {
    if (!m) FR_FAIL(FR_ERR_INVALID, "model is NULL");
    if (m->n_tables <= 0 || m->n_segments <= 0 || !m->tables || !m->segments)
        FR_FAIL(FR_ERR_INVALID, "model '%s': empty table or segment list", m->name);
    if (m->record_len <= 0 || m->record_len % 4) FR_FAIL(FR_ERR_INVALID, "record_len %d must be a positive multiple of 4", m->record_len);
    if (m->dense_len < 0 || m->dense_len % 4) FR_FAIL(FR_ERR_INVALID, "dense_len %d must be a multiple of 4", m->dense_len);
    if (m->fc[0] != m->record_len) FR_FAIL(FR_ERR_INVALID, "fc[0]=%d must equal record_len=%d", m->fc[0], m->record_len);
    for (int i = 1; i < 5; i++)
        if (m->fc[i] <= 0) FR_FAIL(FR_ERR_INVALID, "fc[%d]=%d must be positive", i, m->fc[i]);
    for (int i = 0; i < 4; i++)
        if (m->fc[i] % 8) FR_FAIL(FR_ERR_INVALID, "fc[%d]=%d must be a multiple of 8 (operands move in groups of 8 k)", i, m->fc[i]);
    for (int i = 1; i < 4; i++)
        if (m->fc[i] % 32) FR_FAIL(FR_ERR_INVALID, "fc[%d]=%d must be a multiple of 32 (32x32 MFMA output tiles)", i, m->fc[i]);
    if (m->fc[4] != 1) FR_FAIL(FR_ERR_INVALID, "only OUTPUT_FEATURE_LEN == 1 is supported (got %d)", m->fc[4]);
    if (m->layout != FR_LAYOUT_SEMANTIC && m->layout != FR_LAYOUT_BLOCKED) FR_FAIL(FR_ERR_INVALID, "bad layout %d", m->layout);
    if (m->index_mode != FR_INDEX_PER_TABLE && m->index_mode != FR_INDEX_PER_ITEM && m->index_mode != FR_INDEX_PER_BANK) FR_FAIL(FR_ERR_INVALID, "bad index_mode %d", m->index_mode);
    for (int t = 0; t < m->n_tables; t++) {
        const fr_table_desc &d = m->tables[t];
        if (d.dim <= 0 || d.dim % 4 || d.dim > 1024) FR_FAIL(FR_ERR_INVALID, "table %d: dim %d must be a multiple of 4 in (0,1024]", t, d.dim);
        if (d.rows <= 0 || d.rows > 0xFFFFFFFFLL) FR_FAIL(FR_ERR_INVALID, "table %d: rows %lld out of range", t, (long long)d.rows);
        if (d.mem_class < 0 || d.mem_class > 2 || d.table_id < 0 || d.table_id > 255 || d.source < 0 || d.source > 1)
            FR_FAIL(FR_ERR_INVALID, "table %d: bad class/id/source", t);
    }
    // segments must tile [0, record_len) in order; every table must own exactly one TABLE segment
    int pos = 0, dense_seen = 0;
    std::vector<int> owner(m->n_tables, 0);
    int last_source = -1;
    std::vector<int> source_seen;
    for (int s = 0; s < m->n_segments; s++) {
        const fr_segment &g = m->segments[s];
        if (g.rec_offset != pos) FR_FAIL(FR_ERR_INVALID, "segment %d starts at %d, expected %d", s, g.rec_offset, pos);
        if (g.len <= 0 || g.len % 4 || g.src_col < 0 || g.src_col % 4) FR_FAIL(FR_ERR_INVALID, "segment %d: len/src_col must be multiples of 4", s);
        if (g.source < 0 || g.source > 2) FR_FAIL(FR_ERR_INVALID, "segment %d: bad source %d", s, g.source);
        if (g.source != last_source) {  // each source must be one contiguous run (needed by the BLOCKED layout)
            for (int prev : source_seen)
                if (prev == g.source) FR_FAIL(FR_ERR_INVALID, "segment %d: source %d is not contiguous in the record", s, g.source);
            source_seen.push_back(g.source);
            last_source = g.source;
        }
        if (g.kind == FR_SEG_DENSE) {
            if (g.src_col + g.len > m->dense_len) FR_FAIL(FR_ERR_INVALID, "segment %d exceeds dense_len", s);
            dense_seen += g.len;
        } else if (g.kind == FR_SEG_TABLE || g.kind == FR_SEG_COPY) {
            if (g.src < 0 || g.src >= m->n_tables) FR_FAIL(FR_ERR_INVALID, "segment %d: table %d out of range", s, g.src);
            const fr_table_desc &d = m->tables[g.src];
            if (g.src_col + g.len > d.dim) FR_FAIL(FR_ERR_INVALID, "segment %d exceeds table %d's row", s, g.src);
            if (g.kind == FR_SEG_TABLE) {
                if (g.src_col != 0 || g.len != d.dim) FR_FAIL(FR_ERR_INVALID, "segment %d: TABLE segment must be a whole row", s);
                owner[g.src]++;
            }
        } else {
            FR_FAIL(FR_ERR_INVALID, "segment %d: bad kind %d", s, g.kind);
        }
        pos += g.len;
    }
    if (pos != m->record_len) FR_FAIL(FR_ERR_INVALID, "segments cover %d floats, record_len is %d", pos, m->record_len);
    if (dense_seen != m->dense_len) FR_FAIL(FR_ERR_INVALID, "dense segments cover %d floats, dense_len is %d", dense_seen, m->dense_len);
    for (int t = 0; t < m->n_tables; t++)
        if (owner[t] != 1) FR_FAIL(FR_ERR_INVALID, "table %d appears in %d TABLE segments (must be exactly 1)", t, owner[t]);
    return FR_OK;
}

extern "C" int fr_model_clone_scaled(const fr_model_desc *src, double row_scale, int64_t min_rows, int64_t max_rows,
                                     fr_model_desc **out) {
    if (!out) FR_FAIL(FR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int rc = fr_model_validate(src);
    if (rc) return rc;
    if (!(row_scale > 0.0) || min_rows < 1) FR_FAIL(FR_ERR_INVALID, "row_scale must be > 0 and min_rows >= 1");
    // one allocation: desc | tables | segments
    size_t bytes = sizeof(fr_model_desc) + sizeof(fr_table_desc) * src->n_tables + sizeof(fr_segment) * src->n_segments;
    char *mem = (char *)malloc(bytes);
    if (!mem) FR_FAIL(FR_ERR_OOM, "out of host memory");
    fr_model_desc *m = (fr_model_desc *)mem;
    fr_table_desc *t = (fr_table_desc *)(mem + sizeof(fr_model_desc));
    fr_segment *s = (fr_segment *)(mem + sizeof(fr_model_desc) + sizeof(fr_table_desc) * src->n_tables);
    *m = *src;
    memcpy(t, src->tables, sizeof(fr_table_desc) * src->n_tables);
    memcpy(s, src->segments, sizeof(fr_segment) * src->n_segments);
    for (int i = 0; i < src->n_tables; i++) {
        double r = std::floor((double)t[i].rows * row_scale + 0.5);
        int64_t rows = (int64_t)r;
        if (rows < min_rows) rows = min_rows;
        if (max_rows > 0 && rows > max_rows) rows = max_rows;
        if (rows > 0xFFFFFFFFLL) {
            free(mem);
            FR_FAIL(FR_ERR_INVALID, "table %d: scaled row count %lld exceeds 2^32-1", i, (long long)rows);
        }
        t[i].rows = rows;
    }
    m->tables = t;
    m->segments = s;
    *out = m;
    return FR_OK;
}

extern "C" void fr_model_free(fr_model_desc *m) { free(m); }

// Memory banks (the reference's table_HBMx / table_DDRx / table_PLRAMx kernel arguments, embedding_47_krnl.hpp:32-71): bank key =
// (source, mem_class, bank), numbered by first appearance in the table list (tables are in wire order).
int fr_bank_map(const fr_model_desc &m, std::vector<int> &bank_of_table, std::vector<int64_t> &bank_min_rows) {
    bank_of_table.assign(m.n_tables, 0);
    bank_min_rows.clear();
    std::vector<int64_t> keys;
    for (int t = 0; t < m.n_tables; t++) {
        const fr_table_desc &d = m.tables[t];
        const int64_t key = ((int64_t)d.source << 40) | ((int64_t)d.mem_class << 32) | (uint32_t)d.bank;
        int b = -1;
        for (size_t i = 0; i < keys.size(); i++)
            if (keys[i] == key) { b = (int)i; break; }
        if (b < 0) {
            b = (int)keys.size();
            keys.push_back(key);
            bank_min_rows.push_back(d.rows);
        }
        if (d.rows < bank_min_rows[b]) bank_min_rows[b] = d.rows;
        bank_of_table[t] = b;
    }
    return (int)keys.size();
}

extern "C" int fr_model_bank_map(const fr_model_desc *m, int32_t *bank_of_table, int64_t *bank_rows) {
    int rc = fr_model_validate(m);
    if (rc) return rc;
    std::vector<int> bot;
    std::vector<int64_t> rows;
    const int nb = fr_bank_map(*m, bot, rows);
    if (bank_of_table)
        for (int t = 0; t < m->n_tables; t++) bank_of_table[t] = bot[t];
    if (bank_rows)
        for (int b = 0; b < nb; b++) bank_rows[b] = rows[b];
    return nb;
}

extern "C" int fr_model_index_cols(const fr_model_desc *m) {
    if (!m) return FR_ERR_INVALID;
    if (m->index_mode == FR_INDEX_PER_ITEM) return 1;
    if (m->index_mode == FR_INDEX_PER_TABLE) return m->n_tables;
    return fr_model_bank_map(m, nullptr, nullptr);
}

extern "C" int64_t fr_model_table_bytes(const fr_model_desc *m) {
    if (!m || !m->tables) return 0;
    int64_t b = 0;
    for (int i = 0; i < m->n_tables; i++) b += m->tables[i].rows * (int64_t)m->tables[i].dim * 4;
    return b;
}
