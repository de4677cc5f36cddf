// zgpt2_main.cpp — C++ restatement of the reference's host program (src/main.zig) on top of the
// drop-in boundary: GPTConfig / State / MLP / Block / GPT / generate are written against
// ops::{Linear, Embedding, LayerNorm, CausalSelfAttention, gelu} (include/zgpt2_ops.hpp) exactly the
// way main.zig is written against ops.zig — one op call at a time with host buffers the caller
// allocated once — and, with --model-tier, against the device-resident zg_gpt_* tier (one FFI call
// per generation).  There are no GPT-2 checkpoints or vocab files offline, so weights come from the
// portable synthetic generator (bit-identical to zig_gpt2_amd/synth.py), the prompt is a list of
// token ids, and sampling is greedy argmax instead of the reference's time-seeded multinomial.
//
//   zgpt2_main <tiny|tiny3|nano-char|124M> <weight_seed | raw weight directory> <tok,tok,...> <n_steps> [--model-tier]
// prints the tokens after every step on one line (prompt tokens included, main.zig:339-340).
//   zgpt2_main <model> <weights> "<tok,tok,...;tok,...;...>" <n_steps> --gpus N [--plan]
// the multi-GPU case: the ';'-separated prompts are independent units, block-partitioned over N processes, one
// per GPU (rank r on device r); rank 0 loads the weights and ONE RCCL broadcast of the weight arena carries them to the
// others (zg_gpt_broadcast_weights); every rank generates its prompts in lock step; one line per prompt, in prompt order.
// --plan prints the partition and exits without touching a GPU.
#include <poll.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <array>
#include <stdexcept>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/zgpt2_ops.hpp"

using ops::Slice;
typedef std::vector<float> Buf;

// ---------------------------------------------------------------- synthetic weights (synth.py twin)
static inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static Buf fill_normal(uint64_t seed, size_t n, float mean, float std_) {
    const uint64_t key = mix64(seed + 0x9E3779B97F4A7C15ULL);
    const float scale = (float)((double)std_ / 37837.22753904532);
    Buf out(n);
    for (size_t i = 0; i < n; ++i) {
        const uint64_t r = mix64(key + (i + 1) * 0x9E3779B97F4A7C15ULL);
        const int32_t s = (int32_t)(r & 0xFFFF) + (int32_t)((r >> 16) & 0xFFFF) + (int32_t)((r >> 32) & 0xFFFF) +
                          (int32_t)((r >> 48) & 0xFFFF) - 131070;
        float v = (float)s * scale;
        v = v + mean;
        uint32_t b;
        memcpy(&b, &v, 4);
        b += 0x7FFFu + ((b >> 16) & 1u);  // round to bf16 (RNE) so fp32 and bf16 storage agree
        b &= 0xFFFF0000u;
        memcpy(&v, &b, 4);
        out[i] = v;
    }
    return out;
}

// ---------------------------------------------------------------- src/main.zig:5-23
struct GPTConfig {
    size_t vocab_size, context_size, n_layer, n_heads, n_embed;
};

// ---------------------------------------------------------------- src/main.zig:26-65
struct State {
    Buf pos_emb, x, o, logits, _h, _4xh, _qkv, _q, _k, _v, _attn;
    explicit State(const GPTConfig& c)
        : pos_emb(c.n_embed), x(c.n_embed), o(c.n_embed), logits(c.vocab_size), _h(c.n_embed), _4xh(4 * c.n_embed),
          _qkv(3 * c.n_embed), _q(c.n_embed), _k(c.context_size * c.n_embed), _v(c.context_size * c.n_embed),
          _attn(c.context_size) {}
};

struct Weights {  // tensor order / seeds of zig_gpt2_amd/synth.py::tensor_specs
    Buf wte, wpe, ln_f_g, ln_f_b;
    struct Layer {
        Buf t[12];
    };
    std::vector<Layer> h;
    Weights(const GPTConfig& c, uint64_t seed) {
        const size_t E = c.n_embed;
        wte = fill_normal(seed * 4096 + 0, c.vocab_size * E, 0.f, 0.02f);
        wpe = fill_normal(seed * 4096 + 1, c.context_size * E, 0.f, 0.02f);
        ln_f_g = fill_normal(seed * 4096 + 2, E, 1.f, 0.02f);
        ln_f_b = fill_normal(seed * 4096 + 3, E, 0.f, 0.02f);
        const size_t sizes[12] = {E, E, 3 * E * E, 3 * E, E * E, E, E, E, 4 * E * E, 4 * E, 4 * E * E, E};
        const float means[12] = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        h.resize(c.n_layer);
        for (size_t l = 0; l < c.n_layer; ++l)
            for (int s = 0; s < 12; ++s) h[l].t[s] = fill_normal(seed * 4096 + 16 + 16 * l + s, sizes[s], means[s], 0.02f);
    }

    // The reference's raw weight directory (load_linear / load_layer_norm / load_embedding, main.zig:210-269, which
    // hard-code "models/124M/raw/"): one little-endian fp32 file per tensor, `model-<tf name with / -> ->`, Linear
    // weights already [out, in] (download_weights.py:57-64).  Unlike load_tensor (ops.zig:318, short reads ignored) a
    // file of the wrong size is an error.
    static Buf read_tensor(const std::string& dir, const std::string& stem, size_t n) {
        const std::string path = dir + "/model-" + stem;
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open " + path);
        Buf out(n);
        const size_t got = fread(out.data(), sizeof(float), n, f);
        const bool more = got == n && fgetc(f) != EOF;
        fclose(f);
        if (got != n || more) throw std::runtime_error(path + ": size does not match the config");
        return out;
    }
    Weights(const GPTConfig& c, const std::string& dir) {
        const size_t E = c.n_embed;
        wte = read_tensor(dir, "wte", c.vocab_size * E);
        wpe = read_tensor(dir, "wpe", c.context_size * E);
        ln_f_g = read_tensor(dir, "ln_f-g", E);
        ln_f_b = read_tensor(dir, "ln_f-b", E);
        const char* stems[12] = {"ln_1-g", "ln_1-b", "attn-c_attn-w", "attn-c_attn-b", "attn-c_proj-w", "attn-c_proj-b",
                                 "ln_2-g", "ln_2-b", "mlp-c_fc-w", "mlp-c_fc-b", "mlp-c_proj-w", "mlp-c_proj-b"};
        const size_t sizes[12] = {E, E, 3 * E * E, 3 * E, E * E, E, E, E, 4 * E * E, 4 * E, 4 * E * E, E};
        h.resize(c.n_layer);
        for (size_t l = 0; l < c.n_layer; ++l)
            for (int s = 0; s < 12; ++s) h[l].t[s] = read_tensor(dir, "h" + std::to_string(l) + "-" + stems[s], sizes[s]);
    }
};

// ---------------------------------------------------------------- src/main.zig:67-83
struct MLP {
    ops::Linear c_fc, c_proj;
    void forward(Slice<const float> inputs, State& state) const {
        c_fc.forward(inputs, state._4xh);
        ops::gelu(state._4xh);
        c_proj.forward(Slice<const float>(state._4xh.data(), state._4xh.size()), state.o);
    }
};

// ---------------------------------------------------------------- src/main.zig:85-147
struct Block {
    size_t n_embed;
    ops::LayerNorm ln_1;
    ops::CausalSelfAttention attn;
    ops::LayerNorm ln_2;
    MLP mlp;
    Buf k_cache, v_cache;
    void forward(size_t seq_len, State& state) {  // inputs == state.x, as main.zig:187 passes it
        const size_t E = n_embed;
        state._h = state.x;
        ln_1.forward(state._h);
        attn.forward(seq_len, Slice<const float>(state._h.data(), E), Slice<float>(k_cache.data(), seq_len * E),
                     Slice<float>(v_cache.data(), seq_len * E), state.o, state._qkv, state._q,
                     Slice<float>(state._k.data(), seq_len * E), Slice<float>(state._v.data(), seq_len * E),
                     Slice<float>(state._attn.data(), seq_len));
        for (size_t i = 0; i < E; ++i) {  // main.zig:136-139
            state._h[i] = state.o[i] + state.x[i];
            state.x[i] = state._h[i];
        }
        ln_2.forward(state._h);
        mlp.forward(Slice<const float>(state._h.data(), E), state);
        for (size_t i = 0; i < E; ++i) {  // main.zig:142-145
            state.o[i] += state.x[i];
            state.x[i] = state.o[i];
        }
    }
};

// ---------------------------------------------------------------- src/main.zig:149-208
struct GPT {
    GPTConfig config;
    ops::Embedding wte, wpe;
    std::vector<Block> h;
    ops::LayerNorm ln_f;
    ops::Linear lm_head;

    GPT(const GPTConfig& c, const Weights& w) : config(c) {  // load_gpt, main.zig:304-314
        const size_t E = c.n_embed;
        auto S = [](const Buf& b) { return Slice<const float>(b.data(), b.size()); };
        wte = ops::Embedding::init(E, S(w.wte));
        wpe = ops::Embedding::init(E, S(w.wpe));
        ln_f = ops::LayerNorm::init(E, S(w.ln_f_g), S(w.ln_f_b));
        lm_head = ops::Linear::init(E, c.vocab_size, S(w.wte), Slice<const float>());  // main.zig:312
        for (size_t l = 0; l < c.n_layer; ++l) {
            const Buf* t = w.h[l].t;
            Block b;
            b.n_embed = E;
            b.ln_1 = ops::LayerNorm::init(E, S(t[0]), S(t[1]));
            b.attn = ops::CausalSelfAttention::init(c.n_heads, E, ops::Linear::init(E, 3 * E, S(t[2]), S(t[3])),
                                                    ops::Linear::init(E, E, S(t[4]), S(t[5])));
            b.ln_2 = ops::LayerNorm::init(E, S(t[6]), S(t[7]));
            b.mlp = MLP{ops::Linear::init(E, 4 * E, S(t[8]), S(t[9])), ops::Linear::init(4 * E, E, S(t[10]), S(t[11]))};
            b.k_cache.assign(c.context_size * E, 0.f);  // main.zig:298-299
            b.v_cache.assign(c.context_size * E, 0.f);
            h.push_back(std::move(b));
        }
    }

    void forward(size_t seq_len, size_t token, bool compute_logits, State& state) {  // main.zig:178-195
        const size_t pos = seq_len - 1;
        wpe.forward(Slice<const size_t>(&pos, 1), state.pos_emb);
        wte.forward(Slice<const size_t>(&token, 1), state.x);
        for (size_t i = 0; i < config.n_embed; ++i) state.x[i] += state.pos_emb[i];
        for (auto& blk : h) blk.forward(seq_len, state);
        ln_f.forward(state.x);
        if (compute_logits) lm_head.forward(Slice<const float>(state.x.data(), state.x.size()), state.logits);
    }

    size_t sample_greedy(size_t seq_len, size_t token, State& state) {  // replaces main.zig:198-207
        forward(seq_len, token, true, state);
        size_t best = 0;
        for (size_t i = 1; i < state.logits.size(); ++i)
            if (state.logits[i] > state.logits[best]) best = i;
        return best;
    }
};

// ---------------------------------------------------------------- src/main.zig:322-342 (greedy)
static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// (t64: seconds spent on the first 64 positions — BASELINE configs[0]'s window — when the run is at least that long)
static std::vector<size_t> generate(GPT& gpt, const std::vector<size_t>& inputs, size_t n_steps, State& state, double* t64 = nullptr) {
    std::vector<size_t> out;
    size_t token = 0;
    const double t0 = now_s();
    for (size_t s = 0; s < n_steps; ++s) {
        if (s < inputs.size()) {
            token = inputs[s];
            gpt.forward(s + 1, token, false, state);
        } else {
            token = gpt.sample_greedy(s + 1, token, state);
        }
        out.push_back(token);
        if (s + 1 == 64 && t64) *t64 = now_s() - t0;
    }
    return out;
}

// How a device-resident handle stores the matrices: the synthetic weights are bf16-representable (bf16 storage is lossless),
// a checkpoint read from a raw directory is ordinary fp32 (download_weights.py:57-64) — rounding it to bf16 moves the logits by
// 6e-3 of their scale and flips 2 of 64 greedy picks at 124M (tests/test_weight_storage_gpu.py), outside the 1e-3 bound — so it
// keeps the reference's fp32.  Decided from the SOURCE, not the values: every rank of a multi-GPU run must build the same arena.
static unsigned weight_flags(bool from_dir) { return from_dir ? ZG_GPT_WEIGHTS_F32 : ZG_GPT_WEIGHTS_BF16; }

static std::vector<size_t> generate_model_tier(const GPTConfig& c, const Weights& w, const std::vector<size_t>& inputs,
                                               size_t n_steps, bool from_dir) {
    zg_gpt_config cfg{c.vocab_size, c.context_size, c.n_layer, c.n_heads, c.n_embed};
    zg_gpt* g = nullptr;
    ops::check(zg_gpt_create(&g, &cfg, 1, weight_flags(from_dir)));
    const Buf* top[4] = {&w.wte, &w.wpe, &w.ln_f_g, &w.ln_f_b};
    for (int s = 0; s < 4; ++s) ops::check(zg_gpt_load_tensor(g, s, top[s]->data(), top[s]->size()));
    for (size_t l = 0; l < c.n_layer; ++l)
        for (int s = 0; s < 12; ++s) ops::check(zg_gpt_load_block_tensor(g, l, s, w.h[l].t[s].data(), w.h[l].t[s].size()));
    std::vector<size_t> out(n_steps);
    const size_t len = inputs.size();
    ops::check(zg_gpt_generate_greedy(g, inputs.data(), inputs.size(), &len, n_steps, out.data(), out.size()));
    zg_gpt_destroy(g);
    return out;
}

// ---------------------------------------------------------------- multi-GPU: prompts sharded, weights broadcast once
// prompt indices of `rank`: contiguous blocks whose sizes differ by at most one (zig_gpt2_amd/shard.py shard_prompts)
static void shard_prompts(size_t n_prompts, int world, int rank, size_t& begin, size_t& count) {
    const size_t base = n_prompts / world, extra = n_prompts % world;
    begin = rank * base + std::min<size_t>(rank, extra);
    count = base + ((size_t)rank < extra ? 1 : 0);
}

static bool read_all(int fd, void* buf, size_t n) {
    char* p = static_cast<char*>(buf);
    while (n > 0) {
        const ssize_t r = read(fd, p, n);
        if (r <= 0) return false;
        p += r;
        n -= (size_t)r;
    }
    return true;
}

// one rank: its device, the communicator, the handle, the weights (loaded on rank 0, received elsewhere), its prompts
static int run_rank(int rank, int world, const GPTConfig& c, const std::string& wsrc, bool from_dir, uint64_t seed,
                    const std::vector<std::vector<size_t>>& prompts, size_t n_steps, int id_in_fd, const std::vector<int>& id_out_fds, int out_fd) {
    try {
        // (ZGPT2_ALL_RANKS_ON_DEVICE=d: test hook — every rank on device d, with tests/stub_rccl as the transport, so that the
        // rank plumbing of this program can run with N > 1 on a one-GPU box)
        const char* one_dev = getenv("ZGPT2_ALL_RANKS_ON_DEVICE");
        ops::check(zg_init(one_dev ? atoi(one_dev) : rank));
        unsigned char id[ZG_DIST_ID_BYTES];
        if (rank == 0) {
            ops::check(zg_dist_unique_id(id, sizeof id));
            for (int fd : id_out_fds)
                if (write(fd, id, sizeof id) != (ssize_t)sizeof id) throw std::runtime_error("id pipe");
        } else if (!read_all(id_in_fd, id, sizeof id)) {
            throw std::runtime_error("rank 0 sent no communicator id");
        }
        ops::check(zg_dist_init(id, sizeof id, rank, world));
        size_t begin, count;
        shard_prompts(prompts.size(), world, rank, begin, count);
        // every rank creates the same handle shape (the weight region of the arena does not depend on the batch)
        zg_gpt_config cfg{c.vocab_size, c.context_size, c.n_layer, c.n_heads, c.n_embed};
        zg_gpt* g = nullptr;
        ops::check(zg_gpt_create(&g, &cfg, std::max<size_t>(count, 1), weight_flags(from_dir)));
        if (rank == 0) {  // load_gpt (main.zig:304-314) on one GPU only
            const Weights w = from_dir ? Weights(c, wsrc) : Weights(c, seed);
            const Buf* top[4] = {&w.wte, &w.wpe, &w.ln_f_g, &w.ln_f_b};
            for (int s = 0; s < 4; ++s) ops::check(zg_gpt_load_tensor(g, s, top[s]->data(), top[s]->size()));
            for (size_t l = 0; l < c.n_layer; ++l)
                for (int s = 0; s < 12; ++s) ops::check(zg_gpt_load_block_tensor(g, l, s, w.h[l].t[s].data(), w.h[l].t[s].size()));
        }
        float ms = 0.f;
        ops::check(zg_gpt_broadcast_weights(g, 0, &ms));
        if (rank == 0) fprintf(stderr, "weights broadcast to %d rank(s) in %.3f ms\n", world, ms);
        std::vector<size_t> out(count * n_steps);
        if (count > 0) {
            size_t stride = 0;
            for (size_t i = 0; i < count; ++i) stride = std::max(stride, prompts[begin + i].size());
            std::vector<size_t> flat(count * stride, 0), lens(count);
            for (size_t i = 0; i < count; ++i) {
                lens[i] = prompts[begin + i].size();
                std::copy(prompts[begin + i].begin(), prompts[begin + i].end(), flat.begin() + i * stride);
            }
            ops::check(zg_gpt_generate_greedy(g, flat.data(), stride, lens.data(), n_steps, out.data(), out.size()));
        }
        zg_gpt_destroy(g);
        ops::check(zg_dist_finalize());
        if (!out.empty() && write(out_fd, out.data(), out.size() * sizeof(size_t)) != (ssize_t)(out.size() * sizeof(size_t)))
            throw std::runtime_error("result pipe");
    } catch (const std::exception& e) {
        fprintf(stderr, "rank %d: error: %s\n", rank, e.what());
        return 1;
    }
    return 0;
}

static int main_multi_gpu(int world, bool plan_only, const GPTConfig& c, const std::string& wsrc, bool from_dir, uint64_t seed,
                          const std::vector<std::vector<size_t>>& prompts, size_t n_steps) {
    if (plan_only) {
        for (int r = 0; r < world; ++r) {
            size_t begin, count;
            shard_prompts(prompts.size(), world, r, begin, count);
            printf("rank %d:", r);
            for (size_t i = 0; i < count; ++i) printf(" %zu", begin + i);
            printf("\n");
        }
        return 0;
    }
    // The ranks are started BEFORE anything touches a GPU (a process that has initialised HIP must not fork workers).  Pipes:
    // rank 0 -> every other rank (the 128-byte communicator id), every rank -> this process (its token rows).
    std::vector<int> id_rd(world, -1), id_wr, out_rd(world, -1);
    std::vector<pid_t> pids(world);
    std::vector<std::array<int, 2>> idp(world), outp(world);
    for (int r = 0; r < world; ++r)
        if ((r > 0 && pipe(idp[r].data()) != 0) || pipe(outp[r].data()) != 0) return 1;
    for (int r = 1; r < world; ++r) id_wr.push_back(idp[r][1]);
    for (int r = 0; r < world; ++r) {
        pids[r] = fork();
        if (pids[r] < 0) return 1;
        if (pids[r] == 0) {
            for (int q = 0; q < world; ++q) close(outp[q][0]);
            const int rc = run_rank(r, world, c, wsrc, from_dir, seed, prompts, n_steps, r > 0 ? idp[r][0] : -1, r == 0 ? id_wr : std::vector<int>(),
                                    outp[r][1]);
            _exit(rc);
        }
    }
    for (int r = 0; r < world; ++r) {
        close(outp[r][1]);
        if (r > 0) {
            close(idp[r][0]);
            close(idp[r][1]);
        }
    }
    // Collect every rank's token rows as they arrive (poll: a rank's rows may exceed a pipe buffer) while watching for a rank that
    // ends badly — no such device, unreadable weights ...: the others would wait for it in the communicator's rendezvous or in the
    // broadcast for ever, so they are ended here, by their pids, and the run fails.
    std::vector<std::vector<char>> got(world);
    std::vector<size_t> want(world);
    std::vector<bool> exited(world, false), open_fd(world, true);
    for (int r = 0; r < world; ++r) {
        size_t begin, count;
        shard_prompts(prompts.size(), world, r, begin, count);
        want[r] = count * n_steps * sizeof(size_t);
    }
    int failed = 0, live = world;
    while (live > 0 && failed == 0) {
        std::vector<pollfd> fds;
        std::vector<int> who;
        for (int r = 0; r < world; ++r)
            if (open_fd[r]) {
                fds.push_back(pollfd{outp[r][0], POLLIN, 0});
                who.push_back(r);
            }
        if (!fds.empty()) poll(fds.data(), fds.size(), 100);
        for (size_t i = 0; i < fds.size(); ++i)
            if (fds[i].revents & (POLLIN | POLLHUP)) {
                char buf[65536];
                const ssize_t n = read(fds[i].fd, buf, sizeof buf);
                if (n > 0) got[who[i]].insert(got[who[i]].end(), buf, buf + n);
                else open_fd[who[i]] = false;
            }
        for (int r = 0; r < world; ++r) {
            if (exited[r]) continue;
            int status = 0;
            if (waitpid(pids[r], &status, WNOHANG) == pids[r]) {
                exited[r] = true;
                --live;
                if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) ++failed;
            }
        }
    }
    if (failed) {
        for (int r = 0; r < world; ++r)
            if (!exited[r]) {
                kill(pids[r], SIGKILL);
                waitpid(pids[r], nullptr, 0);
            }
        fprintf(stderr, "a rank failed: the run is abandoned\n");
        return 1;
    }
    for (int r = 0; r < world; ++r) {  // (every rank has exited: drain what is left in its pipe)
        char buf[65536];
        ssize_t n;
        while (open_fd[r] && (n = read(outp[r][0], buf, sizeof buf)) > 0) got[r].insert(got[r].end(), buf, buf + n);
        if (got[r].size() != want[r]) {
            fprintf(stderr, "rank %d returned %zu of %zu bytes\n", r, got[r].size(), want[r]);
            return 1;
        }
        const size_t* rows = reinterpret_cast<const size_t*>(got[r].data());
        for (size_t i = 0; i < want[r] / sizeof(size_t) / (n_steps ? n_steps : 1) && n_steps; ++i)
            for (size_t st = 0; st < n_steps; ++st) printf("%zu%s", rows[i * n_steps + st], st + 1 < n_steps ? " " : "\n");
    }
    return failed ? 1 : 0;
}

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s <tiny|tiny3|nano-char|124M> <seed | raw weight dir> <tok,tok,...> <n_steps> [--model-tier]\n", argv[0]);
        return 2;
    }
    const std::string name = argv[1];
    GPTConfig config;
    if (name == "tiny") config = {257, 64, 2, 2, 128};
    else if (name == "tiny3") config = {131, 48, 3, 3, 192};
    else if (name == "nano-char") config = {65, 256, 6, 6, 384};
    else if (name == "124M") config = {50257, 1024, 12, 12, 768};  // main.zig:346
    else {
        fprintf(stderr, "unknown model '%s' (tiny | tiny3 | nano-char | 124M)\n", name.c_str());
        return 2;
    }
    const std::string wsrc = argv[2];  // digits: seed of the synthetic weights; anything else: a raw weight directory
    const bool from_dir = wsrc.find_first_not_of("0123456789") != std::string::npos;
    const uint64_t seed = from_dir ? 0 : strtoull(argv[2], nullptr, 10);
    auto number = [](const char* s, size_t* out) {  // a non-empty string of digits
        if (*s == '\0' || std::string(s).find_first_not_of("0123456789") != std::string::npos) return false;
        *out = strtoull(s, nullptr, 10);
        return true;
    };
    size_t n_steps = 0;
    if (!number(argv[4], &n_steps) || n_steps > config.context_size) {
        fprintf(stderr, "n_steps '%s': a number of decode steps up to the context size %zu\n", argv[4], config.context_size);
        return 2;
    }
    int gpus = 0;
    bool plan_only = false, model_tier = false;
    for (int i = 5; i < argc; ++i) {
        size_t v = 0;
        if (std::string(argv[i]) == "--gpus" && i + 1 < argc && number(argv[i + 1], &v) && v >= 1 && v <= 64) gpus = (int)v, ++i;
        else if (std::string(argv[i]) == "--plan") plan_only = true;
        else if (std::string(argv[i]) == "--model-tier") model_tier = true;
        else {
            fprintf(stderr, "unknown or incomplete option '%s' (--model-tier | --gpus N [--plan], N = 1..64)\n", argv[i]);
            return 2;
        }
    }
    auto tokens_of = [&](std::string item, std::vector<size_t>* out) {  // "tok,tok,...": ids below the vocabulary size
        for (char* p = strtok(item.data(), ","); p; p = strtok(nullptr, ",")) {
            size_t t = 0;
            if (!number(p, &t) || t >= config.vocab_size) {
                fprintf(stderr, "token '%s': ids are numbers below the vocabulary size %zu\n", p, config.vocab_size);
                return false;
            }
            out->push_back(t);
        }
        if (out->empty()) fprintf(stderr, "empty prompt\n");
        return !out->empty();
    };
    if (gpus > 0) {
        std::vector<std::vector<size_t>> prompts;
        std::string all = argv[3];
        size_t pos = 0;
        while (pos <= all.size()) {
            const size_t end = std::min(all.find(';', pos), all.size());
            std::vector<size_t> one;
            if (!tokens_of(all.substr(pos, end - pos), &one)) return 2;
            prompts.push_back(one);
            pos = end + 1;
        }
        return main_multi_gpu(gpus, plan_only, config, wsrc, from_dir, seed, prompts, n_steps);
    }
    std::vector<size_t> inputs;
    if (!tokens_of(argv[3], &inputs)) return 2;
    try {
        ops::check(zg_init(0));
        const Weights w = from_dir ? Weights(config, wsrc) : Weights(config, seed);
        std::vector<size_t> out;
        double t64 = 0.0, t_all = 0.0;
        if (model_tier) {
            const double t0 = now_s();
            out = generate_model_tier(config, w, inputs, n_steps, from_dir);
            t_all = now_s() - t0;  // (handle creation, graph capture and weight upload included: one call does it all here)
        } else {
            State state(config);
            GPT gpt(config, w);
            const double t0 = now_s();
            out = generate(gpt, inputs, n_steps, state, &t64);
            t_all = now_s() - t0;
        }
        // timing on stderr (stdout carries the tokens): what an unchanged main.zig over this ops.zig would see
        fprintf(stderr, "{\"tier\": \"%s\", \"model\": \"%s\", \"steps\": %zu, \"seconds\": %.4f, \"tokens_per_s\": %.1f, \"first_64_tokens_per_s\": %.1f}\n",
                model_tier ? "model" : "op", name.c_str(), n_steps, t_all, t_all > 0 ? n_steps / t_all : 0.0, t64 > 0 ? 64.0 / t64 : 0.0);
        for (size_t i = 0; i < out.size(); ++i) printf("%zu%s", out[i], i + 1 < out.size() ? " " : "\n");
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
