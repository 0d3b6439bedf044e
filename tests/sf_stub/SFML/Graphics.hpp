// Declarations only -- NOT SFML and not a build of anything: just enough of the `sf::` names that
// PathTracerDemo.cpp's main() mentions for `g++ -fsyntax-only` to type-check that file against this repository's
// headers (tests/test_demo_main_syntax.py).  Nothing here is ever linked or run; the window, texture and PNG export
// stay out of scope.  The shapes follow SFML 2's public API as the demo uses it (PathTracerDemo.cpp:524-545, 912-1090).
#pragma once
#include <cstdint>
#include <string>

namespace sf {

struct Vector2u {
    unsigned int x, y;
};

class Color {
public:
    Color(std::uint8_t r, std::uint8_t g, std::uint8_t b, std::uint8_t a = 255);
};

class Image {
public:
    Image();
    void create(unsigned int width, unsigned int height);
    Vector2u getSize() const;
    void setPixel(unsigned int x, unsigned int y, const Color&);
    bool saveToFile(const std::string& path) const;
};

class Texture {
public:
    Texture();
    bool create(unsigned int width, unsigned int height);
    bool loadFromImage(const Image&);
};

class Sprite {
public:
    explicit Sprite(const Texture&);
};

struct FloatRect {
    FloatRect(float left, float top, float width, float height);
};

class View {
public:
    explicit View(const FloatRect&);
};

class VideoMode {
public:
    VideoMode(unsigned int width, unsigned int height, unsigned int bitsPerPixel = 32);
};

namespace Keyboard {
enum Key { A, D, S, W, Q, E, Dash, Equal, Enter };
}

struct Event {
    enum EventType { Closed, Resized, KeyPressed };
    struct SizeEvent {
        unsigned int width, height;
    };
    struct KeyEvent {
        Keyboard::Key code;
        bool alt, control, shift, system;
    };
    EventType type;
    union {
        SizeEvent size;
        KeyEvent key;
    };
};

class RenderWindow {
public:
    RenderWindow(VideoMode, const std::string& title);
    bool pollEvent(Event&);
    void setView(const View&);
    void clear();
    void draw(const Sprite&);
    void display();
};

}   // namespace sf
