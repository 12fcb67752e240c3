// enumerate -- walks one posting list through the reference's document_enumerator API (block_posting_list.hpp:84-186:
// docid(), freq(), next(), next_geq(), move(), position(), size(); exhausted <=> docid() == num_docs()) as presented by
// ds2i_hip::gpu_index::operator[]. Prints "docid freq" per posting for `next` mode, or the landing (docid, position) of
// every probe for `next_geq` / `move` mode. Used by the tests to hold the adaptor's enumerator against the raw lists.
//
//   enumerate <index_type> <index_file> <term> next
//   enumerate <index_type> <index_file> <term> next_geq <lower_bound>...
//   enumerate <index_type> <index_file> <term> move <position>...
//   enumerate <index_type> <index_file> <replicas> set_recovery   (self-check of set_query<> over N replicas on device 0: a batch
//                                                                  holding an out-of-range term fails, the next batch on the
//                                                                  SAME operator object is answered, equal to one replica's answer)
#include <cstdlib>

#include "../include/ds2i_hip.hpp"
#include "tool_util.hpp"

int main(int argc, const char** argv) {
    if (argc < 5) {
        std::cerr << "usage: " << argv[0] << " <index_type> <index_file> <term> next|next_geq|move [args...]\n";
        return 1;
    }
    const int kind = tool::kind_of(argv[1]);
    if (kind < 0) {
        tool::logger(std::string("ERROR: Unknown type ") + argv[1]);
        return 0;
    }
    try {
        tool::mapped_file m(argv[2]);
        if (std::string(argv[4]) == "set_recovery") {
            const size_t parts = std::max<size_t>(1, std::strtoull(argv[3], nullptr, 10));
            ds2i_hip::gpu_index_set set(kind, m.data, m.size, nullptr, 0, std::vector<int>(parts, 0));
            std::vector<ds2i_hip::term_id_vec> good;
            for (uint32_t q = 0; q < 1200; ++q) good.push_back({q % (uint32_t)set.size(), (q * 7 + 3) % (uint32_t)set.size()});
            std::vector<ds2i_hip::term_id_vec> bad = good;
            bad[bad.size() / 2].push_back((uint32_t)set.size() + 5); // term id out of range: DS2I_ETERM for its ticket
            ds2i_hip::set_query<DS2I_OP_AND> op;
            op.prefer_latency(true); // (1200 queries: fine tickets, so that all replicas' pipelines are in play)
            bool threw = false;
            try { op(set, bad); } catch (std::exception const&) { threw = true; }
            const std::vector<uint64_t> got = op(set, good); // (poisoned pipelines would say DS2I_EBUSY here)
            ds2i_hip::gpu_query_op<DS2I_OP_AND> one;
            const std::vector<uint64_t> want = one(set.replica(0), good);
            std::printf("threw %d equal %d n %zu\n", threw ? 1 : 0, got == want ? 1 : 0, got.size());
            return (threw && got == want) ? 0 : 3;
        }
        ds2i_hip::gpu_index index(kind, m.data, m.size);
        auto e = index[(size_t)std::strtoull(argv[3], nullptr, 10)];
        const std::string mode = argv[4];
        std::printf("size %llu num_docs %llu\n", (unsigned long long)e.size(), (unsigned long long)index.num_docs());
        if (mode == "next") {
            for (; e.docid() < index.num_docs(); e.next())
                std::printf("%llu %llu\n", (unsigned long long)e.docid(), (unsigned long long)e.freq());
        } else if (mode == "next_geq") {
            for (int i = 5; i < argc; ++i) { // forward-only, like the reference (block_posting_list.hpp:124-126)
                e.next_geq(std::strtoull(argv[i], nullptr, 10));
                std::printf("%llu %llu\n", (unsigned long long)e.docid(), (unsigned long long)e.position());
            }
        } else if (mode == "move") {
            for (int i = 5; i < argc; ++i) {
                e.move(std::strtoull(argv[i], nullptr, 10));
                std::printf("%llu %llu\n", (unsigned long long)e.docid(), (unsigned long long)e.position());
            }
            e.reset();
            std::printf("%llu %llu\n", (unsigned long long)e.docid(), (unsigned long long)e.position());
        }
    } catch (std::exception const& ex) {
        tool::logger(std::string("ERROR: ") + ex.what());
        return 2;
    }
    return 0;
}
