// text_cursor.cpp -- see text_cursor.h for the contract and reference citations.
#include "text_cursor.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace ct {

static const size_t kRefWindow = 4096;   // FB_BUFFER_SZ, fileBuffer.h:9

TextCursor::TextCursor(const std::string &path) : path_(path), data_(std::make_shared<std::vector<char>>()) {
    std::vector<char> &data_ = *this->data_;
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    data_.resize(n > 0 ? (size_t)n : 0);
    size_t got = data_.empty() ? 0 : std::fread(data_.data(), 1, data_.size(), f);
    std::fclose(f);
    ok_ = (got == data_.size());
}

void TextCursor::fatal(const std::string &what) const {
    // The reference asserts on malformed input (fileBuffer.cpp:89,144,157,193,196,239);
    // an assert failure is an abort, and so is this.
    std::fprintf(stderr, "cobbletrace: %s: %s (byte %zu)\n", path_.c_str(), what.c_str(), pos_);
    std::abort();
}

char TextCursor::peek() const {
    const std::vector<char> &data_ = *this->data_;
    const size_t n = data_.size();
    if (pos_ < n) return data_[pos_];
    // One position past the end is still "not EOF" in the reference; its window
    // then shows the byte one window earlier (see header).
    if (pos_ == n && n >= kRefWindow) return data_[n - kRefWindow];
    return 0;
}

void TextCursor::skip_space() {
    while (!at_end() && is_space(peek())) advance();
}

void TextCursor::skip_line() {
    while (!at_end() && peek() != '\n' && peek() != '\r') advance();
    skip_space();
}

char TextCursor::token() {
    if (at_end()) return 0;
    skip_space();
    char c = peek();
    advance();
    return c;
}

void TextCursor::expect(char c) {
    char got = token();
    if (got != c) {
        char msg[64];
        std::snprintf(msg, sizeof msg, "expected %c got %c", c, got);
        fatal(msg);
    }
}

std::string TextCursor::quoted() {
    std::string out;
    skip_space();
    if (at_end()) return out;
    if (peek() != '"') fatal("expected a quoted string");
    advance();
    if (at_end()) return out;
    while (out.size() < 1024 && !at_end() && peek() != '"') {   // FB_LINE_SZ, fileBuffer.h:10
        out.push_back(peek());
        advance();
    }
    advance();   // closing quote
    return out;
}

std::string TextCursor::bare_word() {
    std::string out;
    skip_space();
    while (out.size() < 1024 && !at_end() && !is_space(peek())) {
        out.push_back(peek());
        advance();
    }
    return out;
}

bool TextCursor::boolean() {
    std::string w;
    skip_space();
    while (!at_end() && w.size() < 5 && peek() >= 'a' && peek() <= 'z') {
        w.push_back(peek());
        advance();
    }
    if (w == "true") return true;
    if (w == "false") return false;
    fatal("expected true or false");
}

// Decimal text -> binary32, reproducing fileBuffer.cpp:160-207 operation by
// operation: the integer part is Horner-accumulated in float, each fractional
// digit is divided by a float power of ten and added, and only the exponent is
// applied in double (float * pow(10.0, e), rounded back to float).  A leading
// '+' or an exponent '+' is not part of a number, as in the reference.
float TextCursor::number() {
    char c = token();
    float sign = 1.0f, acc = 0.0f, place = 10.0f;
    bool in_fraction = false, in_exponent = false;
    int exponent = 0, exponent_sign = 1;

    if (c == '-') sign = -1.0f;
    else unget();

    while (!at_end()) {
        c = peek();
        if (c >= '0' && c <= '9') {
            const int digit = c - '0';
            if (in_exponent) {
                if (exponent < 100000000) exponent = exponent * 10 + digit;   // (beyond: 10^e is inf or 0 whatever follows; no int overflow)
            } else if (in_fraction) {
                float term = (float)digit;
                term = term / place;
                acc += term;
                place *= 10.0f;
            } else {
                acc = acc * 10.0f;
                acc += (float)digit;
            }
        } else if (c == '.') {
            if (in_fraction) fatal("second decimal point in a number");
            in_fraction = true;
        } else if (c == 'e') {
            if (in_exponent) fatal("second exponent in a number");
            in_exponent = true;
        } else if (c == '-' && in_exponent) {
            exponent_sign = -1;
        } else {
            break;
        }
        advance();
    }
    const float mantissa = sign * acc;
    const double scaled = (double)mantissa * std::pow(10.0, (double)(exponent_sign * exponent));
    return (float)scaled;
}

void TextCursor::vec3_bracketed(double out[3]) {
    expect('[');
    out[0] = number();
    expect(',');
    out[1] = number();
    expect(',');
    out[2] = number();
    expect(']');
}

void TextCursor::vec3_bare(double out[3]) {
    skip_space();
    out[0] = number();
    skip_space();
    out[1] = number();
    skip_space();
    out[2] = number();
}

}  // namespace ct
