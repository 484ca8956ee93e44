// Baseline JPEG -> 8-bit RGB (Jpeg.cpp).  Returns false for anything it does not decode (progressive, arithmetic, CMYK, 12-bit).
#pragma once
#include <vector>
bool load_jpeg(const std::vector<unsigned char>& file, int& w, int& h, std::vector<unsigned char>& rgb);
