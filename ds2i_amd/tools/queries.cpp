// queries -- drop-in counterpart of the reference benchmark driver (queries.cpp:65-153):
//
//   queries <index_type> <query_op[:query_op...]> <index_file> [<wand_data_file>] < query_log
//
// Same argv, same query-log format (one query per line, whitespace separated term ids, queries.hpp:15-27),
// same stats_line JSON keys (type, query, avg, q50, q90, q95 -- microseconds) plus qps / kernel_ms / gpus.
// Where the reference times one query at a time on one core (op_perftest, queries.cpp:13-62), this driver
// sends the whole log through the batched C ABI: 1 untimed + 2 timed passes; "avg" = batch time / queries.
// Quantiles come from a latency pass that submits (a sample of) the queries one per call.
#include <algorithm>
#include <numeric>

#include "../include/ds2i_hip.hpp"
#include "tool_util.hpp"

using namespace ds2i_hip;

template <class Op>
void op_perftest(gpu_index const& index, Op&& op, std::vector<term_id_vec> const& queries, std::string const& type,
                 std::string const& query_type, size_t runs) {
    if (queries.empty()) { // nothing to time (the reference would divide by zero here, queries.cpp:36-40)
        tool::logger("---- " + type + " " + query_type + ": empty query log, nothing to do");
        return;
    }
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
    auto quantile = [&](size_t pct) { return lat[std::min(lat.size() - 1, pct * lat.size() / 100)]; };
    const double q50 = quantile(50), q90 = quantile(90), q95 = quantile(95);
    std::ostringstream os;
    os << "---- " << type << " " << query_type << "\nMean: " << avg << "\n50% quantile: " << q50
       << "\n90% quantile: " << q90 << "\n95% quantile: " << q95;
    tool::logger(os.str());
    std::printf("{\"type\": \"%s\", \"query\": \"%s\", \"avg\": %g, \"q50\": %g, \"q90\": %g, \"q95\": %g, "
                "\"qps\": %g, \"kernel_ms\": %g, \"gpus\": 1}\n",
                type.c_str(), query_type.c_str(), avg, q50, q90, q95, 1e6 / avg, kernel_ms / runs);
}

int main(int argc, const char** argv) {
    if (argc < 4) {
        std::cerr << "usage: " << argv[0] << " <index_type> <query_op[:op...]> <index_file> [<wand_file>] < queries\n";
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
        gpu_index index(kind, m.data, m.size, md ? md->data : nullptr, md ? md->size : 0, 0);
        tool::logger("Performing " + type + " queries");
        std::stringstream ss(query_type);
        std::string t;
        while (std::getline(ss, t, ':')) {
            tool::logger("Query type: " + t);
            if (t == "and") op_perftest(index, and_query(), queries, type, t, 2);
            else if (t == "and_freq") op_perftest(index, and_freq_query(), queries, type, t, 2);
            else if (t == "or") op_perftest(index, or_query(), queries, type, t, 2);
            else if (t == "or_freq") op_perftest(index, or_freq_query(), queries, type, t, 2);
            else if (t == "wand" && md) op_perftest(index, wand_query(10), queries, type, t, 2);
            else if (t == "ranked_and" && md) op_perftest(index, ranked_and_query(10), queries, type, t, 2);
            else if (t == "maxscore" && md) op_perftest(index, maxscore_query(10), queries, type, t, 2);
            else if (t == "ranked_or" && md) op_perftest(index, ranked_or_query(10), queries, type, t, 2);
            else tool::logger("Unsupported query type: " + t);
        }
    } catch (std::exception const& e) {
        tool::logger(std::string("ERROR: ") + e.what());
        return 2;
    }
    return 0;
}
