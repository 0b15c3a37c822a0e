// cpx_host.cpp -- host-side helpers of the file-fed path that are too slow in Python at device speed: the JSON text
// of track positions (hundreds of Region.meta_dictionary() entries per file) as json.dump writes them.  No device code.
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "cpx.h"

namespace {

struct Writer {
  char* out;
  long cap, n;
  void put(const char* s, long len) {
    if (n + len <= cap) memcpy(out + n, s, (size_t)len);
    n += len;
  }
  void lit(const char* s) { put(s, (long)strlen(s)); }
  void spaces(int k) {
    static const char sp[] = "                                                                ";
    while (k > 0) {
      const int m = k < 64 ? k : 64;
      put(sp, m);
      k -= m;
    }
  }
  void integer(long v) {
    char b[24];
    const auto r = std::to_chars(b, b + sizeof(b), v);
    put(b, (long)(r.ptr - b));
  }
  // repr(float) of CPython: shortest digits that round-trip; fixed notation for 1e-4 <= |v| < 1e16, else scientific
  void real(double v) {
    char b[40];
    if (v == 0.0) {
      lit(std::signbit(v) ? "-0.0" : "0.0");
      return;
    }
    if (std::isnan(v)) {
      lit("NaN");
      return;
    }
    if (std::isinf(v)) {
      lit(v > 0 ? "Infinity" : "-Infinity");
      return;
    }
    const double a = std::fabs(v);
    if (a >= 1e-4 && a < 1e16) {
      const auto r = std::to_chars(b, b + sizeof(b), v, std::chars_format::fixed);
      long len = (long)(r.ptr - b);
      bool dot = false;
      for (long i = 0; i < len; ++i) dot |= b[i] == '.';
      put(b, len);
      if (!dot) lit(".0");
    } else {
      const auto r = std::to_chars(b, b + sizeof(b), v, std::chars_format::scientific);
      put(b, (long)(r.ptr - b));
    }
  }
};

// Region.meta_dictionary (reference src/ml_tools/rectangle.py:164-177 through track/region.py): x, y, width, height,
// mass, frame_number, pixel_variance = round(value, 2) -- a numpy float32 for a seen region (numpy rounds it in
// float32: rint(v * 100) / 100; the JSON encoder then prints the float64 of that float32), the int 0 for a blank one
// -- blank, in_trap.
void region_dict(Writer& w, const cpx_region& r, int indent, int depth) {
  const bool blank = (r.flags & 1) != 0;
  const char* keys[9] = {"\"x\": ", "\"y\": ", "\"width\": ", "\"height\": ", "\"mass\": ", "\"frame_number\": ",
                         "\"pixel_variance\": ", "\"blank\": ", "\"in_trap\": "};
  const long ints[6] = {r.x, r.y, r.width, r.height, r.mass, r.frame_number};
  w.lit("{");
  for (int k = 0; k < 9; ++k) {
    if (indent > 0) {
      w.lit("\n");
      w.spaces(indent * depth);
    }
    w.lit(keys[k]);
    if (k < 6) {
      w.integer(ints[k]);
    } else if (k == 6) {
      if (blank) {
        w.lit("0");
      } else {
        const float v = rintf(r.pixel_variance * 100.0f) / 100.0f;
        w.real((double)v);
      }
    } else if (k == 7) {
      w.lit(blank ? "true" : "false");
    } else {
      w.lit("false");
    }
    if (k < 8) w.lit(indent > 0 ? "," : ", ");
  }
  if (indent > 0) {
    w.lit("\n");
    w.spaces(indent * (depth - 1));
  }
  w.lit("}");
}

}  // namespace

extern "C" long cpx_format_regions(const cpx_region* regs, int n, long stride_bytes, int indent, int depth, int as_list,
                                   char* out, long cap) {
  if (!regs || n < 0 || !out || cap < 0 || (!as_list && n != 1)) return -1;
  Writer w{out, cap, 0};
  const char* base = reinterpret_cast<const char*>(regs);
  if (!as_list) {
    region_dict(w, *regs, indent, depth);
  } else if (n == 0) {
    w.lit("[]");
  } else {
    w.lit("[");
    for (int i = 0; i < n; ++i) {
      if (indent > 0) {
        w.lit("\n");
        w.spaces(indent * depth);
      }
      region_dict(w, *reinterpret_cast<const cpx_region*>(base + (long)i * stride_bytes), indent, depth + 1);
      if (i + 1 < n) w.lit(indent > 0 ? "," : ", ");
    }
    if (indent > 0) {
      w.lit("\n");
      w.spaces(indent * (depth - 1));
    }
    w.lit("]");
  }
  return w.n <= cap ? w.n : -w.n;
}

// json.dumps(obj) [separators ", " and ": "] -> the text json.dumps(obj, indent=indent) writes: the C encoder of CPython
// only runs without an indent (the indented form goes through the pure-Python encoder, ~1 us per token), so the
// file-fed path encodes compactly and lays the text out here.  Containers open a level, "," ends a line, empty
// containers stay "[]" / "{}"; string contents are copied untouched.
extern "C" long cpx_json_indent(const char* in, long n, int indent, int depth0, char* out, long cap) {
  if (!in || n < 0 || !out || cap < 0 || indent < 1) return -1;
  Writer w{out, cap, 0};
  int depth = depth0;
  bool in_string = false;
  for (long i = 0; i < n; ++i) {
    const char c = in[i];
    if (in_string) {
      w.put(&c, 1);
      if (c == '\\' && i + 1 < n) {
        ++i;
        w.put(in + i, 1);
      } else if (c == '"') {
        in_string = false;
      }
      continue;
    }
    if (c == '"') {
      in_string = true;
      w.put(&c, 1);
    } else if (c == '{' || c == '[') {
      if (i + 1 < n && in[i + 1] == (c == '{' ? '}' : ']')) {
        w.put(in + i, 2);
        ++i;
      } else {
        w.put(&c, 1);
        ++depth;
        w.lit("\n");
        w.spaces(indent * depth);
      }
    } else if (c == '}' || c == ']') {
      --depth;
      w.lit("\n");
      w.spaces(indent * depth);
      w.put(&c, 1);
    } else if (c == ',') {
      w.put(&c, 1);
      w.lit("\n");
      w.spaces(indent * depth);
      if (i + 1 < n && in[i + 1] == ' ') ++i;
    } else {
      w.put(&c, 1);
    }
  }
  return w.n <= cap ? w.n : -w.n;
}
