// Host driver, part 4 of 5: continuous batching (sv_cb_*): requests are rows ("slots") of one decode loop.
#include "engine_internal.h"

// ------------------------------------------------------------------------------------------------
// C ABI: continuous batching.  The reference's worker admits up to 5 concurrent requests
// (serve/model_worker.py:161-172,216-229) and runs each as its own HF generate; here they share ONE decode loop: every
// row ("slot") of the batch is a request with its own sampling parameters, budget, EOS and stop sequence, requests join
// (prefill into free slots while the others keep their KV pages) and leave at any step, and the captured decode step is kept
// per row bucket.  A request produces the same tokens as when it runs alone through sv_generate.
// ------------------------------------------------------------------------------------------------
static int cb_bucket(const sv_engine* e) {
    int hi = 0;
    for (int s2 = 0; s2 < e->cfg.max_batch; ++s2) if (e->cb_used[s2]) hi = s2 + 1;
    int b = 8;
    while (b < hi) b <<= 1;
    return b > e->cfg.max_batch ? e->cfg.max_batch : b;
}

static void cb_step_args(sv_engine* e, CbStepArgs& a, const int32_t* map) {
    a.logits = e->logits; a.ld = e->Vpad; a.V = e->cfg.vocab; a.slots = e->cb_slots; a.slot_map = map;
    a.cur_tok = e->cur_tok; a.positions = e->positions; a.out_tokens = e->out_tok; a.ld_out = e->out_ld;
    a.seen = e->seen; a.seen_words = e->seen_words; a.n_live = e->cb_nlive; a.events = e->cb_events; a.bad = e->d_bad;
}

static int cb_begin(sv_engine* e, hipStream_t st) {
    if (e->cb_active) return 0;
    e->free_pages.clear();
    for (int p = e->num_pages - 1; p >= 0; --p) e->free_pages.push_back(p);
    std::fill(e->cb_used.begin(), e->cb_used.end(), 0);
    for (auto& v : e->cb_pages) v.clear();
    const size_t R = (size_t)e->MT * 32;
    std::vector<int32_t> table((size_t)e->cfg.max_batch * e->pages_per_seq, e->trash_page);
    HIPCHECK(hipMemcpyAsync(e->block_table, table.data(), table.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemsetAsync(e->cb_slots, 0, R * sizeof(CbSlot), st));
    HIPCHECK(hipMemsetAsync(e->positions, 0, R * sizeof(int32_t), st));
    HIPCHECK(hipMemsetAsync(e->cur_tok, 0, R * sizeof(int32_t), st));
    HIPCHECK(hipMemsetAsync(e->cb_nlive, 0, sizeof(int32_t), st));
    HIPCHECK(hipMemsetAsync(e->cb_events, 0, sizeof(int32_t), st));
    HIPCHECK(hipStreamSynchronize(st));                  // `table` is a host temporary
    e->cb_active = true;
    e->cached_B = 0;
    return 0;
}

extern "C" int sv_cb_admit(sv_engine* e, const void* dev_embeds, int32_t n, int32_t S0, const sv_cb_request* reqs,
                           int32_t* slots_out, sv_stream stream) {
    SVCHECK(check_ready(e));
    if (!dev_embeds || !reqs || !slots_out || n < 1) return fail(SV_EINVAL, "sv_cb_admit: null argument or empty batch");
    const sv_config& c = e->cfg;
    if (n > c.max_batch) return fail(SV_EINVAL, "sv_cb_admit: %d requests exceed max_batch %d", n, c.max_batch);
    if (S0 < 1) return fail(SV_EINVAL, "sv_cb_admit: bad prompt length %d", S0);
    for (int i = 0; i < n; ++i) {
        const sv_cb_request& r = reqs[i];
        if (r.max_new_tokens < 1 || S0 + r.max_new_tokens > c.max_seq_len)
            return fail(SV_EINVAL, "sv_cb_admit: request %d: prompt %d + max_new_tokens %d out of range (max_seq_len %d)", i, S0, r.max_new_tokens, c.max_seq_len);
        if (r.n_stop < 0 || r.n_stop > SV_CB_MAXSTOP) return fail(SV_EINVAL, "sv_cb_admit: request %d: stop sequence length %d unsupported (0..%d)", i, r.n_stop, SV_CB_MAXSTOP);
        if (r.do_sample && !(r.temperature > 0.f && r.top_p > 0.f)) return fail(SV_EINVAL, "sv_cb_admit: request %d: temperature and top_p must be > 0", i);
    }
    std::lock_guard<std::mutex> lk(e->mu);
    HIPCHECK(hipSetDevice(c.device));
    HIPCHECK(hipEventRecord(e->gen_event, (hipStream_t)stream));
    hipStream_t st = e->gen_stream;
    HIPCHECK(hipStreamWaitEvent(st, e->gen_event, 0));
    SVCHECK(cb_begin(e, st));
    // free slots (lowest first: keeps the row bucket of the decode graph small) and pages for the whole budget
    std::vector<int> slots;
    size_t need_pages = 0;
    for (int s2 = 0; s2 < c.max_batch && (int)slots.size() < n; ++s2) if (!e->cb_used[s2]) slots.push_back(s2);
    for (int i = 0; i < n; ++i) need_pages += (size_t)(S0 + reqs[i].max_new_tokens + SV_PAGE_TOKENS - 1) / SV_PAGE_TOKENS;
    if ((int)slots.size() < n || need_pages > e->free_pages.size())
        return fail(SV_EBUSY, "sv_cb_admit: %d requests need %d slots / %zu KV pages, %zu / %zu are free (release finished slots first)",
                    n, n, need_pages, slots.size(), e->free_pages.size());
    std::vector<int32_t> rows((size_t)n * e->pages_per_seq, e->trash_page);
    std::vector<CbSlot> hs(n);
    std::vector<int32_t> map(n), pos(n, S0 - 1);
    const bool any_pen = [&] { for (int i = 0; i < n; ++i) if (reqs[i].repetition_penalty > 0.f && reqs[i].repetition_penalty != 1.0f) return true; return false; }();
    for (int i = 0; i < n; ++i) {
        const int s2 = slots[i];
        const sv_cb_request& r = reqs[i];
        const int need = (S0 + r.max_new_tokens + SV_PAGE_TOKENS - 1) / SV_PAGE_TOKENS;
        e->cb_pages[s2].clear();
        for (int k = 0; k < need; ++k) {
            rows[(size_t)i * e->pages_per_seq + k] = e->free_pages.back();
            e->cb_pages[s2].push_back(e->free_pages.back());
            e->free_pages.pop_back();
        }
        e->cb_used[s2] = 1;
        CbSlot& h = hs[i];
        memset(&h, 0, sizeof(h));
        h.live = 1; h.step = 0; h.budget = r.max_new_tokens; h.do_sample = r.do_sample ? 1 : 0; h.temperature = r.temperature;
        h.top_p = r.top_p; h.top_k = r.top_k; h.eos = r.eos_token_id; h.pad = r.pad_token_id; h.min_new = r.min_new_tokens;
        h.penalty = r.repetition_penalty > 0.f ? r.repetition_penalty : 1.0f; h.n_stop = r.n_stop; h.seed = r.seed;
        for (int k = 0; k < r.n_stop; ++k) h.stop[k] = r.stop_ids[k];
        map[i] = s2;
        slots_out[i] = s2;
    }
    // Device side.  Any failure below rolls the host bookkeeping back (slots, pages) and parks the slots' device state, so a
    // failed admit leaks nothing and the caller may simply retry: the slots it was told about are NOT in use on error.
    bool nlive_added = false;
    const int rc = [&]() -> int {
    for (int i = 0; i < n; ++i) {
        const int s2 = slots[i];
        HIPCHECK(hipMemcpyAsync(e->block_table + (size_t)s2 * e->pages_per_seq, rows.data() + (size_t)i * e->pages_per_seq,
                                e->pages_per_seq * sizeof(int32_t), hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(e->cb_slots + s2, &hs[i], sizeof(CbSlot), hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(e->positions + s2, &pos[i], sizeof(int32_t), hipMemcpyHostToDevice, st));
        if (any_pen) HIPCHECK(hipMemsetAsync(e->seen + (size_t)s2 * e->seen_words, 0, e->seen_words * sizeof(uint32_t), st));
    }
    HIPCHECK(hipMemcpyAsync(e->cb_table_pf, rows.data(), rows.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(e->cb_map, map.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    add_i32(e->cb_nlive, n, 1, st);
    nlive_added = true;
    // prompt pass of the NEW requests only (their pages through cb_table_pf); the live slots keep decoding afterwards
    SVCHECK(prefill_forward(e, (const bf16_t*)dev_embeds, n, S0, st, 0, nullptr, e->cb_table_pf));
    CbStepArgs a;
    cb_step_args(e, a, e->cb_map);
    launch_cb_step(a, n, st);                              // first token of every new request, from the prefill logits
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(st));                    // the staging vectors above are host temporaries
    return 0;
    }();
    if (rc) {
        const std::string why = g_err;                      // keep the first error's text
        (void)hipStreamSynchronize(st);
        (void)hipGetLastError();
        std::vector<int32_t> trash(e->pages_per_seq, e->trash_page);
        for (int i = n - 1; i >= 0; --i) {                  // pages go back in reverse order: the free list is as it was
            const int s2 = slots[i];
            for (size_t k = e->cb_pages[s2].size(); k-- > 0;) e->free_pages.push_back(e->cb_pages[s2][k]);
            e->cb_pages[s2].clear();
            e->cb_used[s2] = 0;
            slots_out[i] = -1;
            // best effort on the device: the slot is dead and its block-table row points at the trash page again
            (void)hipMemsetAsync(e->cb_slots + s2, 0, sizeof(CbSlot), st);
            (void)hipMemcpyAsync(e->block_table + (size_t)s2 * e->pages_per_seq, trash.data(), trash.size() * sizeof(int32_t),
                                 hipMemcpyHostToDevice, st);
        }
        if (nlive_added) add_i32(e->cb_nlive, -n, 1, st);
        (void)hipStreamSynchronize(st);
        (void)hipGetLastError();
        g_err = why;
        return rc;
    }
    return 0;
}

extern "C" int sv_cb_step(sv_engine* e, int32_t n_steps, int32_t* n_live_out, sv_stream stream) {
    SVCHECK(check_ready(e));
    if (n_steps < 1 || !n_live_out) return fail(SV_EINVAL, "sv_cb_step: bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->cb_active) return fail(SV_ESTATE, "sv_cb_step: no continuous batch (sv_cb_admit first)");
    HIPCHECK(hipSetDevice(e->cfg.device));
    hipStream_t st = e->gen_stream;
    const int Bb = cb_bucket(e);
    hipGraphExec_t gexec = nullptr;
    int per_launch = 1;
    if (getenv("SV_NO_GRAPH") == nullptr) {
        // one graph per (bucket, steps per launch), kept for the life of the engine: every argument is engine-owned.  The scheduler asks for the same
        // n_steps call after call, so the whole call is ONE graph of n_steps copies of the step (engine_generate.hip: the GPU idles 8.6 us between two
        // graph launches and not at all between two kernels of one graph); key = bucket + 1024 * copies (0: the one-step graph).
        auto capture = [&](int copies, hipGraphExec_t* out) -> int {
            const int key = Bb + 1024 * (copies > 1 ? copies : 0);
            auto it = e->cb_graphs.find(key);
            if (it != e->cb_graphs.end()) { *out = it->second.second; return 0; }
            hipGraph_t g = nullptr;
            hipGraphExec_t ge = nullptr;
            hipError_t ce = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
            if (ce == hipSuccess) {
                for (int u = 0; u < copies; ++u) {
                    decode_forward(e, Bb, st);
                    CbStepArgs a;
                    cb_step_args(e, a, nullptr);
                    launch_cb_step(a, Bb, st);
                }
                ce = hipStreamEndCapture(st, &g);
                if (ce == hipSuccess && g) ce = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            }
            if (ce != hipSuccess) {
                (void)hipGetLastError();
                if (ge) (void)hipGraphExecDestroy(ge);
                if (g) (void)hipGraphDestroy(g);
                if (getenv("SV_REQUIRE_GRAPH")) return fail(SV_EHIP, "hipGraph capture failed: %s", hipGetErrorString(ce));
                *out = nullptr;
                return 0;
            }
            e->cb_graphs[key] = {g, ge};
            *out = ge;
            return 0;
        };
        static const int cap = getenv("SV_GRAPH_STEPS") ? atoi(getenv("SV_GRAPH_STEPS")) : 32;
        if (n_steps >= 2 && n_steps <= cap) {
            SVCHECK(capture(n_steps, &gexec));
            if (gexec) per_launch = n_steps;
        }
        if (!gexec) SVCHECK(capture(1, &gexec));
    }
    for (int i = 0; i < n_steps; i += per_launch) {
        if (gexec) {
            HIPCHECK(hipGraphLaunch(gexec, st));
        } else {
            decode_forward(e, Bb, st);
            CbStepArgs a;
            cb_step_args(e, a, nullptr);
            launch_cb_step(a, Bb, st);
        }
    }
    // the live count and the give-up / non-finite flag in one round trip
    HIPCHECK(hipMemcpyAsync(&e->h_flags[3], e->cb_nlive, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipMemcpyAsync(&e->h_flags[4], e->d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    *n_live_out = e->h_flags[3];
    SVCHECK(report_bad_logits(e, st, "sv_cb_step", e->h_flags[4]));
    e->timing_graph = gexec ? (double)per_launch : 0.0;
    return 0;
}

extern "C" int sv_cb_poll(sv_engine* e, int32_t* host_live, int32_t* host_steps, int32_t capacity) {
    if (!e || !host_live || !host_steps) return fail(SV_EINVAL, "sv_cb_poll: null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (capacity < e->cfg.max_batch) return fail(SV_EINVAL, "sv_cb_poll: capacity %d < max_batch %d", capacity, e->cfg.max_batch);
    for (int s2 = 0; s2 < e->cfg.max_batch; ++s2) { host_live[s2] = 0; host_steps[s2] = 0; }
    if (!e->cb_active) return 0;
    HIPCHECK(hipSetDevice(e->cfg.device));
    std::vector<CbSlot> hs(e->cfg.max_batch);
    HIPCHECK(hipMemcpyAsync(hs.data(), e->cb_slots, hs.size() * sizeof(CbSlot), hipMemcpyDeviceToHost, e->gen_stream));
    HIPCHECK(hipStreamSynchronize(e->gen_stream));
    for (int s2 = 0; s2 < e->cfg.max_batch; ++s2)
        if (e->cb_used[s2]) { host_live[s2] = hs[s2].live; host_steps[s2] = hs[s2].step; }
    return 0;
}

extern "C" int sv_cb_read(sv_engine* e, int32_t slot, int32_t first, int32_t count, int64_t* host_tokens) {
    if (!e || !host_tokens) return fail(SV_EINVAL, "sv_cb_read: null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->cb_active || slot < 0 || slot >= e->cfg.max_batch || !e->cb_used[slot]) return fail(SV_EINVAL, "sv_cb_read: slot %d is not in use", slot);
    if (first < 0 || count < 0 || first + count > e->out_ld) return fail(SV_EINVAL, "sv_cb_read: columns [%d, %d) out of range", first, first + count);
    if (count == 0) return 0;
    HIPCHECK(hipSetDevice(e->cfg.device));
    std::vector<int32_t> tmp(count);
    HIPCHECK(hipMemcpyAsync(tmp.data(), e->out_tok + (size_t)slot * e->out_ld + first, (size_t)count * sizeof(int32_t),
                            hipMemcpyDeviceToHost, e->gen_stream));
    HIPCHECK(hipStreamSynchronize(e->gen_stream));
    for (int i = 0; i < count; ++i) host_tokens[i] = tmp[i];
    return 0;
}

static int cb_release_locked(sv_engine* e, int slot, hipStream_t st) {
    // a slot released while still generating is stopped first (live -> 0, live counter adjusted on the host's view)
    CbSlot h;
    HIPCHECK(hipMemcpyAsync(&h, e->cb_slots + slot, sizeof(CbSlot), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    if (h.live) add_i32(e->cb_nlive, -1, 1, st);
    HIPCHECK(hipMemsetAsync(e->cb_slots + slot, 0, sizeof(CbSlot), st));
    HIPCHECK(hipMemsetAsync(e->positions + slot, 0, sizeof(int32_t), st));
    HIPCHECK(hipMemsetAsync(e->cur_tok + slot, 0, sizeof(int32_t), st));
    fill_i32(e->block_table + (size_t)slot * e->pages_per_seq, e->trash_page, e->pages_per_seq, st);
    for (int pg : e->cb_pages[slot]) e->free_pages.push_back(pg);
    e->cb_pages[slot].clear();
    e->cb_used[slot] = 0;
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int sv_cb_release(sv_engine* e, int32_t slot) {
    if (!e) return fail(SV_EINVAL, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->cb_active || slot < 0 || slot >= e->cfg.max_batch || !e->cb_used[slot]) return fail(SV_EINVAL, "sv_cb_release: slot %d is not in use", slot);
    HIPCHECK(hipSetDevice(e->cfg.device));
    return cb_release_locked(e, slot, e->gen_stream);
}

extern "C" int sv_cb_reset(sv_engine* e) {
    if (!e) return fail(SV_EINVAL, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->cb_active) return 0;
    HIPCHECK(hipSetDevice(e->cfg.device));
    for (int s2 = 0; s2 < e->cfg.max_batch; ++s2)
        if (e->cb_used[s2]) SVCHECK(cb_release_locked(e, s2, e->gen_stream));
    HIPCHECK(hipStreamSynchronize(e->gen_stream));
    e->cb_active = false;
    return 0;
}
