// create_freq_index / create_wand_data -- host tools that turn a ds2i binary collection
// (<basename>.docs/.freqs/.sizes, reference README.md:152-174) into the on-disk images the query path loads.
// They stand in for the reference's create_freq_index.cpp:45-110 and create_wand_data.cpp:8-29, which cannot
// be built here (succinct/FastPFor/Boost absent). CPU only.
//
//   create_freq_index <index_type> <collection_basename> <output_index> [<output_wand_data>]
#include "../../include/ds2i_build.h"
#include "tool_util.hpp"

static void write_blob(const char* path, ds2i_blob* b) {
    FILE* f = std::fopen(path, "wb");
    if (!f) throw std::runtime_error(std::string("cannot write ") + path);
    std::fwrite(ds2i_blob_data(b), 1, ds2i_blob_size(b), f);
    std::fclose(f);
}

int main(int argc, const char** argv) {
    if (argc < 4) {
        std::cerr << "usage: " << argv[0] << " <index_type> <collection_basename> <output_index> [<output_wand_data>]\n";
        return 1;
    }
    const int kind = tool::kind_of(argv[1]);
    if (kind < 0) {
        tool::logger(std::string("ERROR: Unknown type ") + argv[1]);
        return 0;
    }
    try {
        const std::string base = argv[2];
        tool::mapped_file fdocs((base + ".docs").c_str()), ffreqs((base + ".freqs").c_str());
        tool::binary_sequences docs(fdocs), freqs(ffreqs);
        const uint32_t* d;
        const uint32_t* f;
        size_t nd, nf;
        if (!docs.next(d, nd) || nd != 1) throw std::invalid_argument("First sequence should only contain number of documents");
        const uint64_t num_docs = d[0];
        ds2i_builder* b = nullptr;
        if (ds2i_builder_create(kind, num_docs, &b)) throw std::runtime_error("ds2i_builder_create failed");
        ds2i_wand_builder* w = nullptr;
        std::unique_ptr<tool::mapped_file> fsizes;
        if (argc > 4) {
            fsizes.reset(new tool::mapped_file((base + ".sizes").c_str()));
            tool::binary_sequences sizes(*fsizes);
            const uint32_t* s;
            size_t ns;
            if (!sizes.next(s, ns) || ns != num_docs) throw std::invalid_argument("sizes file does not match num_docs");
            if (ds2i_wand_create(s, num_docs, &w)) throw std::runtime_error("ds2i_wand_create failed");
        }
        size_t lists = 0, postings = 0;
        while (docs.next(d, nd)) {
            if (!freqs.next(f, nf) || nf != nd) throw std::invalid_argument("docs/freqs sequences out of step");
            if (ds2i_builder_add_posting_list(b, nd, d, f)) throw std::runtime_error("add_posting_list failed");
            if (w && ds2i_wand_add_list(w, nd, d, f)) throw std::runtime_error("wand_add_list failed");
            ++lists;
            postings += nd;
        }
        ds2i_blob* img = nullptr;
        if (ds2i_builder_freeze(b, &img)) throw std::runtime_error("freeze failed");
        write_blob(argv[3], img);
        std::ostringstream os;
        os << lists << " sequences, " << postings << " postings, " << ds2i_blob_size(img) << " bytes ("
           << (8.0 * ds2i_blob_size(img) / postings) << " bits/posting)";
        tool::logger(os.str());
        ds2i_blob_free(img);
        ds2i_builder_free(b);
        if (w) {
            ds2i_blob* wi = nullptr;
            if (ds2i_wand_freeze(w, &wi)) throw std::runtime_error("wand freeze failed");
            write_blob(argv[4], wi);
            ds2i_blob_free(wi);
            ds2i_wand_free(w);
        }
    } catch (std::exception const& e) {
        tool::logger(std::string("ERROR: ") + e.what());
        return 2;
    }
    return 0;
}
