// alp/storer.hpp — sequential byte cursors over caller-owned memory.  API-compatible with the two helper structs the
// reference exposes as alp::storer::MemStorer<DRY> / MemReader (a dry storer only counts).  Both are thin views over
// one shared cursor type; nothing here touches the GPU.
#ifndef ALP_STORER_HPP
#define ALP_STORER_HPP
#include <cstddef>
#include <cstdint>
#include <cstring>

// Typical use (what a column writer built on the per-vector API does):
//     alp::storer::MemStorer<true>  sizer;            // first pass: DRY, just add up the sizes
//     alp::storer::MemStorer<false> writer(buffer);   // second pass: copy bit width, base, packed words, exceptions ...
//     alp::storer::MemReader        reader(buffer);   // read them back in the same order
// The batch path of this repository has its own container (alpgpu_column_to_blob / alpgpu_column_from_blob in alpgpu.h):
// one header followed by the HBM records, so a whole column moves with four copies instead of one call per field.
namespace alp { namespace storer {

namespace detail {
//! position inside a caller-owned byte range; the two public cursors differ only in the direction of the copy
struct byte_cursor {
	uint8_t* origin = nullptr;
	size_t   cursor = 0;

	uint8_t* here() const { return origin + cursor; }
	void     advance(size_t n) { cursor += n; }
};
} // namespace detail

template <bool DRY = false>
struct MemStorer {
	uint8_t* out_buffer    = nullptr;
	size_t   buffer_offset = 0;

	MemStorer() = default;
	explicit MemStorer(uint8_t* destination)
	    : out_buffer(destination) {}

	void   set_buffer(uint8_t* destination) { out_buffer = destination; }
	void   reset() { buffer_offset = 0; }
	size_t get_size() { return buffer_offset; }

	//! append `bytes_to_store` bytes of `in`; in DRY mode only the size is accumulated
	void store(void* in, size_t bytes_to_store) {
		detail::byte_cursor c {out_buffer, buffer_offset};
		if constexpr (!DRY) { std::memcpy(c.here(), in, bytes_to_store); }
		c.advance(bytes_to_store);
		buffer_offset = c.cursor;
	}
};

struct MemReader {
	uint8_t* in_buffer     = nullptr;
	size_t   buffer_offset = 0;

	MemReader() = default;
	explicit MemReader(uint8_t* source)
	    : in_buffer(source) {}

	void   set_buffer(uint8_t* source) { in_buffer = source; }
	void   reset() { buffer_offset = 0; }
	size_t get_size() { return buffer_offset; }

	//! copy the next `bytes_to_read` bytes into `out`
	void read(void* out, size_t bytes_to_read) {
		detail::byte_cursor c {in_buffer, buffer_offset};
		std::memcpy(out, c.here(), bytes_to_read);
		c.advance(bytes_to_read);
		buffer_offset = c.cursor;
	}
};

}} // namespace alp::storer
#endif // ALP_STORER_HPP
