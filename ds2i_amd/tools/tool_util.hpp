// small host utilities shared by the command line tools (file mapping, ds2i binary collections)
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace tool {

// read-only mmap (the reference uses boost::iostreams::mapped_file_source, queries.cpp:76)
struct mapped_file {
    const uint8_t* data = nullptr;
    size_t size = 0;
    explicit mapped_file(const char* path) {
        int fd = ::open(path, O_RDONLY);
        if (fd < 0) throw std::runtime_error(std::string("cannot open ") + path);
        struct stat st;
        ::fstat(fd, &st);
        size = (size_t)st.st_size;
        void* p = size ? ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
        ::close(fd);
        if (size && p == MAP_FAILED) throw std::runtime_error(std::string("cannot mmap ") + path);
        data = (const uint8_t*)p;
    }
    ~mapped_file() { if (data) ::munmap((void*)data, size); }
};

// ds2i binary collection: little-endian u32 stream of [len][len x u32] sequences
// (reference README.md:152-174, binary_collection.hpp:127-142; empty sequences are skipped)
struct binary_sequences {
    const uint32_t* w;
    size_t n, pos = 0;
    binary_sequences(const mapped_file& f) : w((const uint32_t*)f.data), n(f.size / 4) {}
    bool next(const uint32_t*& begin, size_t& len) {
        while (pos < n) {
            size_t l = w[pos++];
            if (!l) continue;
            if (l > n - pos) l = n - pos; // truncated file
            begin = w + pos;
            len = l;
            pos += l;
            return true;
        }
        return false;
    }
};

inline double get_time_usecs() {
    timeval tv;
    gettimeofday(&tv, nullptr);
    return double(tv.tv_sec) * 1000000 + double(tv.tv_usec);
}

inline int kind_of(std::string const& type) {
    // DS2I_INDEX_TYPES (index_types.hpp:41), numbered like enum ds2i_hip_index_kind
    static const char* names[] = {"block_optpfor", "block_varint", "block_interpolative", "block_qmx", "block_mixed",
                                  "opt", "ef", "single", "uniform"};
    for (int i = 0; i < 9; ++i)
        if (type == names[i]) return i;
    return -1;
}

inline void logger(std::string const& msg) { std::cerr << msg << std::endl; }

} // namespace tool
