// text_cursor.h -- in-memory tokenizer used by the scene and mesh loaders.
//
// Behavioural contract (what the reference's windowed reader does on the same
// bytes, fileBuffer.cpp:21-258), restated over a whole-file buffer:
//   * white space is ' ', '\t', '\r', '\n'                      (fileBuffer.cpp:37-44)
//   * "end of input" is position > size, i.e. one position past the last byte is
//     still readable                                            (fileBuffer.cpp:21-24)
//     -- the byte seen there is what the reference's 4 KiB window still holds
//     from the previous block, file[size-4096]; for files shorter than one
//     window it is indeterminate in the reference and reads as NUL here.
//   * numbers are accumulated in binary32 exactly as fileBuffer.cpp:160-207 does
//     (see TextCursor::number).
#ifndef CT_TEXT_CURSOR_H
#define CT_TEXT_CURSOR_H

#include <cstddef>
#include <memory>
#include <string>
#include <vector>

namespace ct {

class TextCursor {
public:
    // Loads the whole file; on failure ok() is false.
    explicit TextCursor(const std::string &path);

    // A second cursor over the SAME bytes (shared, not copied), starting at `position`: what the parallel
    // mesh import gives each worker.
    TextCursor(const TextCursor &whole, size_t position) : path_(whole.path_), data_(whole.data_), pos_(position), ok_(whole.ok_) {}

    bool ok() const { return ok_; }
    size_t position() const { return pos_; }
    void seek(size_t position) { pos_ = position; }
    size_t size() const { return data_->size(); }
    const char *bytes() const { return data_->data(); }
    static bool is_space(char c) { return c == '\n' || c == '\r' || c == ' ' || c == '\t'; }
    const std::string &path() const { return path_; }
    bool at_end() const { return pos_ > data_->size(); }

    void skip_space();
    void skip_line();                    // to end of line, then over white space
    char token();                        // next non-space byte (0 at end), consumed
    void unget() { if (pos_ > 0) --pos_; }
    void expect(char c);                 // token() must equal c, else fatal()
    std::string quoted();                // "..." contents; empty at end of input
    std::string bare_word();             // run of non-space bytes
    bool boolean();                      // true | false, else fatal()
    float number();                      // binary32 accumulate, see .cpp
    void vec3_bracketed(double out[3]);  // [a, b, c]
    void vec3_bare(double out[3]);       // a b c

    // number -> integer as the reference's assignments convert (`int n = GetNumber(&fb)`, objectLoader.cpp:177;
    // `uint32_t totalVertices = GetNumber(&fb)`, :159), defined for every value: in range it is the truncation; outside
    // -- undefined in the language -- what the reference's x86-64 binary does (cvttss2si: INT_MIN; the low half of
    // the 64-bit conversion).  Hostile files reach these with 1e30 and NaN.
    static int to_int(double v) { return (v > -2147483649.0 && v < 2147483648.0) ? (int)v : (-2147483647 - 1); }
    static unsigned to_u32(double v) { return (v > -9223372036854775808.0 && v < 9223372036854775808.0) ? (unsigned)(long long)v : 0u; }

    [[noreturn]] void fatal(const std::string &what) const;

private:
    char peek() const;
    void advance() { ++pos_; }

    std::string path_;
    std::shared_ptr<std::vector<char>> data_;
    size_t pos_ = 0;
    bool ok_ = false;
};

}  // namespace ct
#endif
