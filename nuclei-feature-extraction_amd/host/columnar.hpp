// columnar.hpp -- Parquet and Arrow IPC (feather v2) writers for the feature table, no dependencies.
//
// The reference hands its DataFrame to polars' ParquetWriter / IpcWriter by output extension
// (main.rs:169-180: "parquet" | "pqt", "ipc" | "feather"). The table here is always one string column
// (`centroid`) followed by F numeric columns -- float32, except the columns the reference's frame holds as float64
// (the seven polygon columns of the geometry set, shape.rs:116-127: `wide`) -- which is all these writers support:
//   * Parquet: format version 1, one row group per 131072 rows, one uncompressed PLAIN data page per column
//     chunk, REQUIRED fields (the table has no nulls; NaN stays NaN), `centroid` BYTE_ARRAY annotated UTF8/STRING.
//     Metadata is Thrift compact protocol, written by the small encoder below.
//   * Arrow IPC file: metadata V5, little endian, `centroid` Utf8 (int32 offsets), features FloatingPoint(SINGLE | DOUBLE),
//     one record batch per 131072 rows, no compression, no dictionaries. Messages are FlatBuffers, written by
//     the small back-to-front builder below.
// Both files are read back by pyarrow in tests/test_cli.py. The input is the row-major [n, F] block the engine
// returns; columns are gathered one row group at a time so the extra memory stays at one group.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace nfxhost {

constexpr int64_t kRowGroup = 131072;

struct FeatureTable {
    const std::vector<std::string> *names; // "centroid" + F feature names
    const std::vector<std::string> *keys;  // n centroid keys
    const float *rows;                     // [n, F] row-major
    int64_t n;
    int F;
    // columns the reference emits as f64 (nfx_feature_set_column_dtype): wide[c] = index into wide_rows' columns, -1 for f32
    const int *wide = nullptr;             // [F] or null (all f32)
    const double *wide_rows = nullptr;     // [n, W] row-major
    int W = 0;
    int wide_of(int c) const { return wide ? wide[c] : -1; } // c: feature column, 0-based
};

class FileSink {
  public:
    explicit FileSink(FILE *f) : f_(f) {}
    void write(const void *p, size_t len)
    {
        if (len && fwrite(p, 1, len, f_) != len) throw std::runtime_error("short write to the output file");
        pos_ += (int64_t)len;
    }
    void pad_to(int64_t align)
    {
        static const char zeros[64] = {0};
        const int64_t r = pos_ % align;
        if (r) write(zeros, (size_t)(align - r));
    }
    int64_t pos() const { return pos_; }

  private:
    FILE *f_;
    int64_t pos_ = 0;
};

// ---- Thrift compact protocol (only what parquet.thrift's FileMetaData and PageHeader need) -------------------------
class ThriftCompact {
  public:
    std::string out;
    void begin_struct() { last_.push_back(0); }
    void end_struct() { out.push_back(0); last_.pop_back(); }
    void i32(int id, int32_t v) { field(id, 5); zigzag(v); }
    void i64(int id, int64_t v) { field(id, 6); zigzag(v); }
    void str(int id, const std::string &s) { field(id, 8); bytes(s); }
    void struct_field(int id) { field(id, 12); begin_struct(); }
    void list_field(int id, int elem_type, size_t count) { field(id, 9); list_header(elem_type, count); }
    // list elements
    void elem_i32(int32_t v) { zigzag(v); }
    void elem_str(const std::string &s) { bytes(s); }
    void elem_struct() { begin_struct(); }

  private:
    std::vector<int> last_;
    void varint(uint64_t v)
    {
        while (v >= 0x80) { out.push_back((char)(v | 0x80)); v >>= 7; }
        out.push_back((char)v);
    }
    void zigzag(int64_t v) { varint(((uint64_t)v << 1) ^ (uint64_t)(v >> 63)); }
    void bytes(const std::string &s) { varint(s.size()); out += s; }
    void field(int id, int type)
    {
        const int delta = id - last_.back();
        if (delta > 0 && delta <= 15) out.push_back((char)((delta << 4) | type));
        else { out.push_back((char)type); zigzag(id); }
        last_.back() = id;
    }
    void list_header(int elem_type, size_t count)
    {
        if (count < 15) out.push_back((char)((count << 4) | elem_type));
        else { out.push_back((char)(0xF0 | elem_type)); varint(count); }
    }
};

inline void write_parquet(FILE *f, const FeatureTable &t)
{
    enum { T_FLOAT = 4, T_DOUBLE = 5, T_BYTE_ARRAY = 6, ENC_PLAIN = 0, ENC_RLE = 3, PAGE_DATA = 0, REP_REQUIRED = 0, CONV_UTF8 = 0 };
    struct Chunk { int64_t offset, bytes, values; };
    struct Group { std::vector<Chunk> cols; int64_t rows, bytes; };
    FileSink sink(f);
    sink.write("PAR1", 4);
    const int ncol = t.F + 1;
    std::vector<Group> groups;
    std::string page;
    for (int64_t r0 = 0; r0 < t.n; r0 += kRowGroup) {
        const int64_t rows = std::min(kRowGroup, t.n - r0);
        Group g{{}, rows, 0};
        for (int c = 0; c < ncol; ++c) {
            page.clear();
            if (c == 0) {
                for (int64_t i = 0; i < rows; ++i) {
                    const std::string &k = (*t.keys)[(size_t)(r0 + i)];
                    const uint32_t len = (uint32_t)k.size();
                    page.append(reinterpret_cast<const char *>(&len), 4);
                    page += k;
                }
            } else if (t.wide_of(c - 1) >= 0) {
                page.resize((size_t)rows * 8);
                double *dst = reinterpret_cast<double *>(&page[0]);
                const double *src = t.wide_rows + (size_t)r0 * t.W + t.wide_of(c - 1);
                for (int64_t i = 0; i < rows; ++i) dst[i] = src[(size_t)i * t.W];
            } else {
                page.resize((size_t)rows * 4);
                float *dst = reinterpret_cast<float *>(&page[0]);
                const float *src = t.rows + (size_t)r0 * t.F + (c - 1);
                for (int64_t i = 0; i < rows; ++i) dst[i] = src[(size_t)i * t.F];
            }
            ThriftCompact h; // PageHeader
            h.begin_struct();
            h.i32(1, PAGE_DATA);
            h.i32(2, (int32_t)page.size());
            h.i32(3, (int32_t)page.size());
            h.struct_field(5); // DataPageHeader
            h.i32(1, (int32_t)rows);
            h.i32(2, ENC_PLAIN);
            h.i32(3, ENC_RLE);
            h.i32(4, ENC_RLE);
            h.end_struct();
            h.end_struct();
            const Chunk ch{sink.pos(), (int64_t)(h.out.size() + page.size()), rows};
            sink.write(h.out.data(), h.out.size());
            sink.write(page.data(), page.size());
            g.cols.push_back(ch);
            g.bytes += ch.bytes;
        }
        groups.push_back(std::move(g));
    }
    ThriftCompact m; // FileMetaData
    m.begin_struct();
    m.i32(1, 1);
    m.list_field(2, 12, (size_t)ncol + 1); // schema: root + leaves
    m.elem_struct();
    m.str(4, "root");
    m.i32(5, ncol);
    m.end_struct();
    for (int c = 0; c < ncol; ++c) {
        m.elem_struct();
        m.i32(1, c == 0 ? T_BYTE_ARRAY : (t.wide_of(c - 1) >= 0 ? T_DOUBLE : T_FLOAT));
        m.i32(3, REP_REQUIRED);
        m.str(4, (*t.names)[(size_t)c]);
        if (c == 0) {
            m.i32(6, CONV_UTF8);
            m.struct_field(10); // LogicalType union
            m.struct_field(1);  // STRING: StringType {}
            m.end_struct();
            m.end_struct();
        }
        m.end_struct();
    }
    m.i64(3, t.n);
    m.list_field(4, 12, groups.size());
    for (const Group &g : groups) {
        m.elem_struct();
        m.list_field(1, 12, g.cols.size());
        for (int c = 0; c < ncol; ++c) {
            const Chunk &ch = g.cols[(size_t)c];
            m.elem_struct(); // ColumnChunk
            m.i64(2, ch.offset);
            m.struct_field(3); // ColumnMetaData
            m.i32(1, c == 0 ? T_BYTE_ARRAY : (t.wide_of(c - 1) >= 0 ? T_DOUBLE : T_FLOAT));
            m.list_field(2, 5, 2);
            m.elem_i32(ENC_PLAIN);
            m.elem_i32(ENC_RLE);
            m.list_field(3, 8, 1);
            m.elem_str((*t.names)[(size_t)c]);
            m.i32(4, 0); // UNCOMPRESSED
            m.i64(5, ch.values);
            m.i64(6, ch.bytes);
            m.i64(7, ch.bytes);
            m.i64(9, ch.offset);
            m.end_struct();
            m.end_struct();
        }
        m.i64(2, g.bytes);
        m.i64(3, g.rows);
        m.end_struct();
    }
    m.str(6, "nuclei-feature-extraction (MI355X engine)");
    m.end_struct();
    sink.write(m.out.data(), m.out.size());
    const uint32_t mlen = (uint32_t)m.out.size();
    sink.write(&mlen, 4);
    sink.write("PAR1", 4);
}

// ---- FlatBuffers, back to front (only what Arrow's Message / Schema / RecordBatch / Footer need) --------------------
// Offsets handed around are distances from the END of the buffer, as in the stock builder.
class FlatBuilder {
  public:
    using Off = uint32_t;
    size_t size() const { return buf_.size(); }
    // finished buffer, front first
    std::string finish(Off root)
    {
        align(8, 4);
        push_u32((uint32_t)(size() + 4 - root));
        return std::string(buf_.rbegin(), buf_.rend());
    }
    Off string(const std::string &s)
    {
        align(4, s.size() + 1 + 4);
        push_byte(0);
        for (size_t i = s.size(); i-- > 0;) push_byte((uint8_t)s[i]);
        push_u32((uint32_t)s.size());
        return (Off)size();
    }
    Off offset_vector(const std::vector<Off> &v)
    {
        align(4, 4 * v.size() + 4);
        for (size_t i = v.size(); i-- > 0;) push_u32((uint32_t)(size() + 4 - v[i]));
        push_u32((uint32_t)v.size());
        return (Off)size();
    }
    // vector of structs made of int64 words (FieldNode, Buffer: 2 words; Block: 3 words with the int32 widened in place)
    Off struct_vector(const std::vector<int64_t> &words, size_t words_per_struct)
    {
        align(8, 8 * words.size());
        for (size_t i = words.size(); i-- > 0;) push_u64((uint64_t)words[i]);
        align(4, 4); // length sits right before the 8-aligned data
        push_u32((uint32_t)(words.size() / words_per_struct));
        return (Off)size();
    }
    // tables: start, add fields in any order, end
    void start_table() { fields_.clear(); table_start_ = size(); }
    void add_i16(int id, int16_t v) { align(2, 2); push_u16((uint16_t)v); note(id); }
    void add_u8(int id, uint8_t v) { push_byte(v); note(id); }
    void add_i64(int id, int64_t v) { align(8, 8); push_u64((uint64_t)v); note(id); }
    void add_offset(int id, Off target) { align(4, 4); push_u32((uint32_t)(size() + 4 - target)); note(id); }
    Off end_table()
    {
        align(4, 4);
        push_u32(0); // soffset to the vtable, patched below
        const size_t table = size();
        int max_id = -1;
        for (auto &fl : fields_) max_id = std::max(max_id, fl.first);
        const size_t vt_len = 4 + 2 * (size_t)(max_id + 1);
        std::vector<uint16_t> vt(2 + (size_t)(max_id + 1), 0);
        vt[0] = (uint16_t)vt_len;
        vt[1] = (uint16_t)(table - table_start_);
        for (auto &fl : fields_) vt[2 + (size_t)fl.first] = (uint16_t)(table - fl.second);
        if (vt_len % 4) push_u16(0); // keep the table 4-aligned behind an odd-length vtable
        for (size_t i = vt.size(); i-- > 0;) push_u16(vt[i]);
        const size_t vtable = size();
        const int32_t soff = (int32_t)(vtable - table);
        for (int b = 0; b < 4; ++b) buf_[table - 1 - (size_t)b] = (uint8_t)((uint32_t)soff >> (8 * b));
        return (Off)table;
    }

  private:
    std::vector<uint8_t> buf_; // reversed: buf_[0] is the LAST byte of the finished buffer
    std::vector<std::pair<int, size_t>> fields_;
    size_t table_start_ = 0;
    void note(int id) { fields_.push_back({id, size()}); }
    void push_byte(uint8_t b) { buf_.push_back(b); }
    void push_u16(uint16_t v) { push_byte((uint8_t)(v >> 8)); push_byte((uint8_t)v); }
    void push_u32(uint32_t v) { for (int b = 3; b >= 0; --b) push_byte((uint8_t)(v >> (8 * b))); }
    void push_u64(uint64_t v) { for (int b = 7; b >= 0; --b) push_byte((uint8_t)(v >> (8 * b))); }
    // pad so that after writing `upcoming` more bytes the size is a multiple of `a`
    void align(size_t a, size_t upcoming)
    {
        while ((size() + upcoming) % a) push_byte(0);
    }
};

namespace arrowfb {
enum { V5 = 4, HDR_SCHEMA = 1, HDR_RECORD_BATCH = 3, TYPE_FLOATING_POINT = 3, TYPE_UTF8 = 5, PRECISION_SINGLE = 1, PRECISION_DOUBLE = 2 };

inline FlatBuilder::Off schema(FlatBuilder &b, const FeatureTable &t)
{
    std::vector<FlatBuilder::Off> fields;
    for (int c = 0; c <= t.F; ++c) {
        const FlatBuilder::Off name = b.string((*t.names)[(size_t)c]);
        const FlatBuilder::Off children = b.offset_vector({});
        b.start_table(); // Utf8 {} or FloatingPoint { precision }
        if (c > 0) b.add_i16(0, t.wide_of(c - 1) >= 0 ? PRECISION_DOUBLE : PRECISION_SINGLE);
        const FlatBuilder::Off type = b.end_table();
        b.start_table(); // Field
        b.add_offset(0, name);
        b.add_u8(1, 0); // nullable = false
        b.add_u8(2, c == 0 ? TYPE_UTF8 : TYPE_FLOATING_POINT);
        b.add_offset(3, type);
        b.add_offset(5, children);
        fields.push_back(b.end_table());
    }
    const FlatBuilder::Off fv = b.offset_vector(fields);
    b.start_table(); // Schema
    b.add_i16(0, 0); // little endian
    b.add_offset(1, fv);
    return b.end_table();
}

inline std::string message(FlatBuilder &b, int header_type, FlatBuilder::Off header, int64_t body_length)
{
    b.start_table();
    b.add_i16(0, V5);
    b.add_u8(1, (uint8_t)header_type);
    b.add_offset(2, header);
    b.add_i64(3, body_length);
    return b.finish(b.end_table());
}
} // namespace arrowfb

inline void write_arrow_ipc(FILE *f, const FeatureTable &t)
{
    using namespace arrowfb;
    struct Block { int64_t offset, meta, body; };
    FileSink sink(f);
    sink.write("ARROW1\0\0", 8);
    auto write_message = [&](const std::string &fb) -> int64_t { // returns the metadata length (prefix + padded flatbuffer)
        const int32_t padded = (int32_t)((fb.size() + 7) & ~(size_t)7);
        const uint32_t cont = 0xFFFFFFFFu;
        sink.write(&cont, 4);
        sink.write(&padded, 4);
        sink.write(fb.data(), fb.size());
        sink.pad_to(8);
        return 8 + (int64_t)padded;
    };
    {
        FlatBuilder b;
        const FlatBuilder::Off s = schema(b, t);
        write_message(message(b, HDR_SCHEMA, s, 0));
    }
    std::vector<Block> blocks;
    std::vector<int32_t> offsets;
    std::string chars;
    std::vector<float> column;
    std::vector<double> wide_column;
    for (int64_t r0 = 0; r0 < t.n; r0 += kRowGroup) {
        const int64_t rows = std::min(kRowGroup, t.n - r0);
        offsets.assign((size_t)rows + 1, 0);
        chars.clear();
        for (int64_t i = 0; i < rows; ++i) {
            chars += (*t.keys)[(size_t)(r0 + i)];
            offsets[(size_t)i + 1] = (int32_t)chars.size();
        }
        auto pad8 = [](int64_t v) { return (v + 7) & ~(int64_t)7; };
        // body: [offsets][chars][col 1]...[col F], each 8-aligned; validity buffers are empty (null_count = 0)
        std::vector<int64_t> nodes, buffers;
        int64_t at = 0;
        nodes.insert(nodes.end(), {rows, 0});
        buffers.insert(buffers.end(), {0, 0});
        buffers.insert(buffers.end(), {at, (int64_t)offsets.size() * 4});
        at += pad8((int64_t)offsets.size() * 4);
        buffers.insert(buffers.end(), {at, (int64_t)chars.size()});
        at += pad8((int64_t)chars.size());
        for (int c = 0; c < t.F; ++c) {
            nodes.insert(nodes.end(), {rows, 0});
            buffers.insert(buffers.end(), {at, 0});
            const int64_t width = t.wide_of(c) >= 0 ? 8 : 4;
            buffers.insert(buffers.end(), {at, rows * width});
            at += pad8(rows * width);
        }
        FlatBuilder b;
        const FlatBuilder::Off bv = b.struct_vector(buffers, 2);
        const FlatBuilder::Off nv = b.struct_vector(nodes, 2);
        b.start_table(); // RecordBatch
        b.add_i64(0, rows);
        b.add_offset(1, nv);
        b.add_offset(2, bv);
        const FlatBuilder::Off rb = b.end_table();
        Block blk{sink.pos(), 0, at};
        blk.meta = write_message(message(b, HDR_RECORD_BATCH, rb, at));
        const int64_t body0 = sink.pos();
        sink.write(offsets.data(), offsets.size() * 4);
        sink.pad_to(8);
        sink.write(chars.data(), chars.size());
        sink.pad_to(8);
        column.resize((size_t)rows);
        wide_column.resize((size_t)rows);
        for (int c = 0; c < t.F; ++c) {
            if (t.wide_of(c) >= 0) {
                const double *src = t.wide_rows + (size_t)r0 * t.W + t.wide_of(c);
                for (int64_t i = 0; i < rows; ++i) wide_column[(size_t)i] = src[(size_t)i * t.W];
                sink.write(wide_column.data(), (size_t)rows * 8);
            } else {
                const float *src = t.rows + (size_t)r0 * t.F + c;
                for (int64_t i = 0; i < rows; ++i) column[(size_t)i] = src[(size_t)i * t.F];
                sink.write(column.data(), (size_t)rows * 4);
            }
            sink.pad_to(8);
        }
        if (sink.pos() - body0 != at) throw std::logic_error("arrow body length mismatch");
        blocks.push_back(blk);
    }
    const uint32_t eos[2] = {0xFFFFFFFFu, 0u};
    sink.write(eos, 8);
    FlatBuilder b;
    std::vector<int64_t> words; // Block { offset: long; metaDataLength: int; <pad>; bodyLength: long }
    for (const Block &k : blocks) words.insert(words.end(), {k.offset, (int64_t)(uint32_t)k.meta, k.body});
    const FlatBuilder::Off rbs = b.struct_vector(words, 3);
    const FlatBuilder::Off dicts = b.struct_vector({}, 3);
    const FlatBuilder::Off s = schema(b, t);
    b.start_table(); // Footer
    b.add_i16(0, V5);
    b.add_offset(1, s);
    b.add_offset(2, dicts);
    b.add_offset(3, rbs);
    const std::string footer = b.finish(b.end_table());
    sink.write(footer.data(), footer.size());
    const int32_t flen = (int32_t)footer.size();
    sink.write(&flen, 4);
    sink.write("ARROW1", 6);
}

} // namespace nfxhost
