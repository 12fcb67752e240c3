// queries -- drop-in counterpart of the reference benchmark driver (queries.cpp:65-153):
//
//   queries <index_type> <query_op[:query_op...]> <index_file> [<wand_data_file>] [--gpus N] < query_log
//
// Same argv, same query-log format (one query per line, whitespace separated term ids, queries.hpp:15-27),
// same stats_line JSON keys (type, query, avg, q50, q90, q95 -- microseconds) plus qps / kernel_ms / hbm_gbps / gpus.
// --gpus N: N replicas of the index, one per device (devices are taken round robin, so N may exceed the device count:
// two replicas on one GPU answer like two GPUs would); every batch is cut into N contiguous slices, one host thread
// and one set of streams per replica (ds2i_hip::gpu_index_set).
// Where the reference times one query at a time on one core (op_perftest, queries.cpp:13-62), this driver
// sends the whole log through the batched C ABI: 1 untimed + 2 timed passes; "avg" = batch time / queries.
// Quantiles come from a latency pass that submits (a sample of) the queries one per call.
#include <algorithm>
#include <cstring>
#include <numeric>

#include "../include/ds2i_hip.hpp"
#include "tool_util.hpp"

using namespace ds2i_hip;

static FILE* g_dump = nullptr;

template <class Index, class Op>
void op_perftest(Index const& index, Op&& op, std::vector<term_id_vec> const& queries, std::string const& type,
                 std::string const& query_type, size_t runs, size_t gpus) {
    if (queries.empty()) { // nothing to time (the reference would divide by zero here, queries.cpp:36-40)
        tool::logger("---- " + type + " " + query_type + ": empty query log, nothing to do");
        return;
    }
    // (a log too short to give every replica a throughput-sized ticket is cut fine, so that every replica takes part)
    if (queries.size() < 1024 * gpus) op.prefer_latency(true);
    double total = 0, kernel_ms = 0;
    for (size_t run = 0; run <= runs; ++run) {
        double tick = tool::get_time_usecs();
        op(index, queries);
        double el = tool::get_time_usecs() - tick;
        if (run) { total += el; kernel_ms += op.stats().kernel_ms; }
    }
    const double avg = total / (runs * queries.size());
    // latency mode: one query per call (H2D + kernel + D2H each), on at most 256 evenly spaced queries
    std::vector<double> lat;
    const size_t step = std::max<size_t>(1, queries.size() / 256);
    for (size_t i = 0; i < queries.size(); i += step) {
        double tick = tool::get_time_usecs();
        op(index, queries[i]);
        lat.push_back(tool::get_time_usecs() - tick);
    }
    std::sort(lat.begin(), lat.end());
    // one more pass with the device counters on (like the reference's Profile=true instantiation, they are an option
    // that costs time): bytes the traversal touched, priced as SURVEY.md 8(d) does, over the kernels' own time
    op.collect_counters(true);
    op(index, queries);
    op.collect_counters(false);
    const double hbm_gbps = op.stats().kernel_ms > 0 ? (double)op.stats().algorithmic_bytes / (op.stats().kernel_ms * 1e-3) / 1e9 : 0.0;
    if (g_dump) { // --dump <file>: the answers of that pass, one line per query: operator, result, top-k score bit patterns
        auto const& counts = op(index, queries);
        for (size_t q = 0; q < queries.size(); ++q) {
            std::fprintf(g_dump, "%s %llu", query_type.c_str(), (unsigned long long)counts[q]);
            if (op.ranked())
                for (float v : op.topk_batch()[q]) { uint32_t b; std::memcpy(&b, &v, 4); std::fprintf(g_dump, " %08x", b); }
            std::fprintf(g_dump, "\n");
        }
    }
    auto quantile = [&](size_t pct) { return lat[std::min(lat.size() - 1, pct * lat.size() / 100)]; };
    const double q50 = quantile(50), q90 = quantile(90), q95 = quantile(95);
    std::ostringstream os;
    os << "---- " << type << " " << query_type << "\nMean: " << avg << "\n50% quantile: " << q50
       << "\n90% quantile: " << q90 << "\n95% quantile: " << q95;
    tool::logger(os.str());
    std::printf("{\"type\": \"%s\", \"query\": \"%s\", \"avg\": %g, \"q50\": %g, \"q90\": %g, \"q95\": %g, "
                "\"qps\": %g, \"kernel_ms\": %g, \"hbm_gbps\": %g, \"gpus\": %zu}\n",
                type.c_str(), query_type.c_str(), avg, q50, q90, q95, 1e6 / avg, kernel_ms / runs, hbm_gbps, gpus);
}

int main(int argc, const char** argv) {
    if (argc < 4) {
        std::cerr << "usage: " << argv[0] << " <index_type> <query_op[:op...]> <index_file> [<wand_file>] < queries\n";
        return 1;
    }
    size_t gpus = 1;
    std::vector<const char*> pos; // positional arguments; "--gpus N" may stand anywhere after the program name
    for (int i = 0; i < argc; ++i) {
        if (std::string(argv[i]) == "--gpus" && i + 1 < argc) { gpus = std::max(1, std::atoi(argv[++i])); continue; }
        if (std::string(argv[i]) == "--dump" && i + 1 < argc) { g_dump = std::fopen(argv[++i], "w"); continue; }
        pos.push_back(argv[i]);
    }
    argc = (int)pos.size();
    argv = pos.data();
    if (argc < 4) {
        std::cerr << "usage: queries <index_type> <query_op[:op...]> <index_file> [<wand_file>] [--gpus N] < queries\n";
        return 1;
    }
    const std::string type = argv[1], query_type = argv[2];
    std::vector<term_id_vec> queries;
    std::string line;
    while (std::getline(std::cin, line)) { // read_query, queries.hpp:15-27
        std::istringstream il(line);
        term_id_vec q;
        term_id_type t;
        while (il >> t) q.push_back(t);
        queries.push_back(q);
    }
    const int kind = tool::kind_of(type);
    if (kind < 0) { // queries.cpp:149-151: log, exit code 0
        tool::logger("ERROR: Unknown type " + type);
        return 0;
    }
    try {
        tool::logger(std::string("Loading index from ") + argv[3]);
        tool::mapped_file m(argv[3]);
        std::unique_ptr<tool::mapped_file> md;
        if (argc > 4) md.reset(new tool::mapped_file(argv[4]));
        const int ndev = std::max(1, ds2i_hip_device_count());
        std::vector<int> devices;
        for (size_t i = 0; i < gpus; ++i) devices.push_back((int)(i % (size_t)ndev));
        gpu_index_set index(kind, m.data, m.size, md ? md->data : nullptr, md ? md->size : 0, devices);
        tool::logger("Performing " + type + " queries on " + std::to_string(gpus) + " replica(s), " + std::to_string(ndev) + " device(s)");
        std::stringstream ss(query_type);
        std::string t;
        while (std::getline(ss, t, ':')) {
            tool::logger("Query type: " + t);
            if (t == "and") op_perftest(index, set_query<DS2I_OP_AND>(), queries, type, t, 2, gpus);
            else if (t == "and_freq") op_perftest(index, set_query<DS2I_OP_AND_FREQ>(), queries, type, t, 2, gpus);
            else if (t == "or") op_perftest(index, set_query<DS2I_OP_OR>(), queries, type, t, 2, gpus);
            else if (t == "or_freq") op_perftest(index, set_query<DS2I_OP_OR_FREQ>(), queries, type, t, 2, gpus);
            else if (t == "wand" && md) op_perftest(index, set_query<DS2I_OP_WAND>(10), queries, type, t, 2, gpus);
            else if (t == "ranked_and" && md) op_perftest(index, set_query<DS2I_OP_RANKED_AND>(10), queries, type, t, 2, gpus);
            else if (t == "maxscore" && md) op_perftest(index, set_query<DS2I_OP_MAXSCORE>(10), queries, type, t, 2, gpus);
            else if (t == "ranked_or" && md) op_perftest(index, set_query<DS2I_OP_RANKED_OR>(10), queries, type, t, 2, gpus);
            else tool::logger("Unsupported query type: " + t);
        }
    } catch (std::exception const& e) {
        tool::logger(std::string("ERROR: ") + e.what());
        return 2;
    }
    if (g_dump) std::fclose(g_dump);
    return 0;
}
