// jpeg_decode.hpp — baseline JPEG -> RGB for Scene::LoadSkybox (scene.cpp:329-377); see jpeg_decode.cpp for what is decoded
// and which published decoder it follows bit for bit.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace p3d {

// rgb: width * height * 3 bytes, top row first (as the file stores it).  false + err: not decodable here.
bool jpeg_decode_rgb(const uint8_t* data, size_t size, std::vector<uint8_t>& rgb, uint32_t& width, uint32_t& height, std::string& err);

}  // namespace p3d
