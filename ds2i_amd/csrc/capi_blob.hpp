// owned byte buffer handed across the C ABI by the builders (ds2i_build.h: ds2i_blob_*)
#pragma once
#include <cstdint>
#include <vector>
struct ds2i_blob { std::vector<uint8_t> data; };
