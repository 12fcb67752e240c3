// thread-local "last error" shared by the two C-ABI translation units
#pragma once
int ds2i_set_error(int code, const char* msg);
const char* ds2i_get_error();
