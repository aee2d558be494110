// alp/storer.hpp — byte cursors over caller-owned buffers (same surface as the reference's include/alp/storer.hpp:10-53).
#ifndef ALP_STORER_HPP
#define ALP_STORER_HPP
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace alp { namespace storer {

template <bool DRY = false>
struct MemStorer {
	uint8_t* out_buffer {nullptr};
	size_t   buffer_offset {0};

	MemStorer() = default;
	explicit MemStorer(uint8_t* out)
	    : out_buffer(out) {}
	void   set_buffer(uint8_t* out) { out_buffer = out; }
	void   reset() { buffer_offset = 0; }
	size_t get_size() { return buffer_offset; }
	void   store(void* in, size_t bytes_to_store) {
        if (!DRY) { std::memcpy(out_buffer + buffer_offset, in, bytes_to_store); }
        buffer_offset += bytes_to_store;
	}
};

struct MemReader {
	uint8_t* in_buffer {nullptr};
	size_t   buffer_offset {0};

	MemReader() = default;
	explicit MemReader(uint8_t* in)
	    : in_buffer(in) {}
	void   set_buffer(uint8_t* in) { in_buffer = in; }
	void   reset() { buffer_offset = 0; }
	size_t get_size() { return buffer_offset; }
	void   read(void* out, size_t bytes_to_read) {
        std::memcpy(out, in_buffer + buffer_offset, bytes_to_read);
        buffer_offset += bytes_to_read;
	}
};

}} // namespace alp::storer
#endif
