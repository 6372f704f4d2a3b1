/*
 * oracle/ref_chat.cpp -- TEST INFRASTRUCTURE.  Model-level harness over the REAL reference host.
 *
 * Our own code: links the reference's chatllm.cpp objects (compiled from /root/reference by oracle/Makefile, minus its
 * main.cpp) and drives ModelObject / generate_next_token (src/chat.h:1371-1411, src/models.cpp:1108-1123) with TOKEN IDS,
 * bypassing the tokenizer and the chat template:
 *     ref_chat MODEL.bin NGL THREADS N_DECODE LOGITS.bin ID [ID ...]
 *   NGL: "cpu" (no -ngl: the reference CPU path) or an -ngl spec such as "all" (our libggml-hip.so module, --ggml_dir = exe dir)
 * It feeds the prompt ids as one chunk, then N_DECODE greedy steps (argmax, first maximum, like Sampler greedy
 * src/models.cpp:676-690), prints the generated ids on stdout and writes every step's logits (float32, vocab each)
 * to LOGITS.bin ("-": not written).  With TEACHER=path it reads the ids to feed at each decode step from a text file instead of its own
 * argmax (teacher forcing, so that two runs stay comparable step by step).
 */
#include "chat.h"
#include "backend.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

/* the host's log sink lives in the reference's main.cpp (src/main.cpp:984), which we do not link */
void log_internal(int level, const char * text) { if (level >= 3 || getenv("REF_CHAT_VERBOSE")) fprintf(stderr, "[ref:%d] %s\n", level, text); }

int main(int argc, char ** argv) {
    if (argc < 7) { fprintf(stderr, "usage: %s MODEL NGL THREADS N_DECODE LOGITS.bin ID...\n", argv[0]); return 2; }
    const std::string path = argv[1], ngl = argv[2];
    const int threads = atoi(argv[3]), n_decode = atoi(argv[4]);
    const char * logits_path = argv[5];
    std::vector<int> ids;
    for (int i = 6; i < argc; i++) ids.push_back(atoi(argv[i]));
    std::vector<int> teacher;
    if (const char * t = getenv("TEACHER")) { std::ifstream f(t); int v; while (f >> v) teacher.push_back(v); }
    /* REF_CHAT_CHUNK_AT=k REF_CHAT_CHUNK="id id ...": decode step k feeds its token followed by these ids as ONE chunk (a second turn's prompt over the live cache) */
    int chunk_at = -1; std::vector<int> chunk;
    if (getenv("REF_CHAT_CHUNK_AT") && getenv("REF_CHAT_CHUNK")) { chunk_at = atoi(getenv("REF_CHAT_CHUNK_AT")); std::istringstream f(getenv("REF_CHAT_CHUNK")); int v; while (f >> v) chunk.push_back(v); }

    std::string exe_dir = argv[0];
    const size_t slash = exe_dir.find_last_of('/');
    exe_dir = slash == std::string::npos ? "." : exe_dir.substr(0, slash);
    try {
        chatllm::ComputeManager::init(exe_dir);
        // REF_CHAT_CACHE: --cache_dtype (f16 | q8_0 ...);  REF_CHAT_FA=1: -fa 1 (src/main.cpp:551-556, 974-978)
        chatllm::ModelObject::extra_args args(-1, "", false, threads, 4096, getenv("REF_CHAT_CACHE") ? getenv("REF_CHAT_CACHE") : "f16");
        if (const char * fa = getenv("REF_CHAT_FA")) args.flash_attention = fa;
        if (ngl != "cpu") args.model_n_gpu_layers["any"] = ngl;
        chatllm::ModelObject obj(path, args);
        chatllm::GenerationConfig gen(obj.model->get_max_length(), obj.model->get_max_length(), false, false, 1, 1.0f, 0.0f, threads, "greedy", 0.0f, 1.0f);
        const bool keep = std::string(logits_path) != "-";          // "-": throughput runs, nothing written
        FILE * fo = keep ? fopen(logits_path, "wb") : nullptr;
        if (keep && !fo) { fprintf(stderr, "cannot open %s\n", logits_path); return 3; }
        std::vector<float> logits;
        std::vector<int> in = ids;
        int n_past = 0;
        if (const char * reps = getenv("REF_CHAT_PREFILL_REPS")) {      // prompt evaluation alone (BASELINE cfg3): one untimed pass, then `reps` timed ones from an empty cache
            const int n = atoi(reps);
            double best = 1e30, sum = 0;
            for (int r = 0; r <= n; r++) {
                obj.model->set_n_past(0);
                const auto p0 = std::chrono::steady_clock::now();
                if (!obj.model->generate_next_token(ids, gen, logits)) { fprintf(stderr, "generate_next_token failed\n"); return 4; }
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - p0).count();
                if (r > 0) { sum += ms; if (ms < best) best = ms; }
            }
            if (n > 0) fprintf(stderr, "prefill: %d tokens, mean %.1f ms, best %.1f ms = %.0f tok/s\n", (int) ids.size(), sum / n, best, ids.size() * 1e3 / best);
        }
        const int skip = n_decode > 40 ? 16 : 0;            // steps left out of the decode timing (first launches, captures, page faults)
        auto t0 = std::chrono::steady_clock::now();
        for (int s = 0; s <= n_decode; s++) {
            if (s == 1 + skip) t0 = std::chrono::steady_clock::now();
            obj.model->set_n_past(n_past);
            if (!obj.model->generate_next_token(in, gen, logits)) { fprintf(stderr, "generate_next_token failed\n"); return 4; }
            n_past += (int) in.size();
            if (fo) fwrite(logits.data(), sizeof(float), logits.size(), fo);
            const int tok = (int)(std::max_element(logits.begin(), logits.end()) - logits.begin());
            printf("%d%s", tok, s == n_decode ? "\n" : " ");
            in.assign(1, s < (int) teacher.size() ? teacher[s] : tok);
            if (s + 1 == chunk_at) in.insert(in.end(), chunk.begin(), chunk.end());      // a second prompt in the middle of the decode: the next graph is a multi-token one
        }
        if (fo) fclose(fo);
        if (n_decode > skip) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            fprintf(stderr, "decode: %d tokens in %.1f ms = %.3f ms/token = %.1f tok/s (context %d -> %d)\n", n_decode - skip, ms, ms / (n_decode - skip),
                    (n_decode - skip) * 1e3 / ms, (int) ids.size() + skip, (int) ids.size() + n_decode);
        }
    } catch (const std::exception & e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
