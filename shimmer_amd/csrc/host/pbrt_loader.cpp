// pbrt_loader.cpp — the PBRT-v4 scene front end behind the C ABI: shm_scene_load_pbrt / shm_scene_parse_pbrt read a
// .pbrt scene description and hand back the ShmSceneDesc (plus the render settings) that the reference's front end would have produced
// for create_integrator. Restates the reference's directive set and defaults (paths relative to /root/reference/src):
//   loading/tokenizer.rs, loading/parser.rs:216-351, 353-875   tokens, directives, parameter lists ("type name" value | [ values ])
//   loading/parser_target.rs:50-184, loading/scene.rs:1221-2033 BasicSceneBuilder: graphics state (CTM, material, area light,
//                                                               reverse orientation), attribute stack, named materials / textures /
//                                                               coordinate systems, object definitions and instances, defaults
//   material.rs:56-110, 259-285, 353-420, 539-575, 688-720, 820-900, 1055-1170, 1297-1310      Material::create and the per-material defaults
//   camera.rs:676-705, 850-890, film.rs:225-330, 482-495, 767-800, sampler.rs:95-99, integrator.rs:16-75
// Scope: the directives the repository's scenes need — LookAt Translate Scale Rotate Identity Transform ConcatTransform CoordinateSystem
// CoordSysTransform ReverseOrientation Camera (perspective / orthographic) Film (rgb) Sampler PixelFilter (box) Integrator Option
// WorldBegin AttributeBegin/End Attribute (per-target default parameters) Material MakeNamedMaterial NamedMaterial Texture (float / spectrum: constant scale mix
// directionmix imagemap) AreaLightSource (diffuse) LightSource (point, infinite: uniform or an environment image) Shape (trianglemesh
// bilinearmesh sphere plymesh) ObjectBegin/End ObjectInstance Include; spectra as "spectrum" samples / named tables / files, "blackbody",
// and "rgb" through the sRGB rgb2spec table (the `.spec` file the reference loads); PNG images (the only format the reference reads,
// host/image_io.hpp) for textures, normal maps and environment lights. What the reference itself leaves todo!() or this backend does not
// take (media, Import, portals, animated transforms, other colour spaces) is reported as SHM_ERR_UNSUPPORTED with the line number, never
// rendered as something else. Host-side only.
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <set>
#include <sstream>

#include "image_io.hpp"
#include "scene_assembly.hpp"

extern "C" __attribute__((visibility("hidden"))) void shm_set_last_error(const char* msg);  // render.hip

namespace pbrt {

// ---- tokens (loading/tokenizer.rs): words, "quoted strings", [ ], # comments -----------------------------------------------------------
struct Token {
    // BAD_QUOTE: an opening quote without its partner — the reference's tokenizer hands the rest of the input over as one token
    // (tokenizer.rs:98-107; tests single_quote, single_quote_with_spaces, just_quote) and its parser rejects it (token.rs is_valid / unquote)
    enum Kind { END, WORD, STRING, LBRACKET, RBRACKET, BAD_QUOTE } kind = END;
    std::string text;  // STRING: without the quotes; BAD_QUOTE: with the opening one
    int line = 0;
    // the token as the reference's Token holds it (quotes kept): what shm_pbrt_tokenize reports
    std::string raw() const { return kind == LBRACKET ? "[" : kind == RBRACKET ? "]" : kind == STRING ? "\"" + text + "\"" : text; }
};
// token.rs:112-170 Directive::from_str — the directive set of parser.rs:373-506 (Token::is_directive)
static bool is_directive_name(const std::string& w) {
    static const std::set<std::string> k = {"Identity", "Translate", "Scale", "Rotate", "LookAt", "CoordinateSystem", "CoordSysTransform", "Transform", "ConcatTransform",
        "TransformTimes", "ActiveTransform", "Include", "Import", "Option", "Camera", "Sampler", "ColorSpace", "Film", "Integrator", "Accelerator", "MakeNamedMedium",
        "MediumInterface", "WorldBegin", "AttributeBegin", "AttributeEnd", "Attribute", "Shape", "ReverseOrientation", "ObjectBegin", "ObjectEnd", "ObjectInstance",
        "LightSource", "AreaLightSource", "Material", "Texture", "MakeNamedMaterial", "NamedMaterial", "PixelFilter"};
    return k.count(w) != 0;
}
class Tokenizer {
public:
    Tokenizer(const std::string& s, const std::string& name) : s_(s), name_(name) {}
    Token next() {
        if (has_peek_) { has_peek_ = false; return peek_; }
        return checked(scan());
    }
    const Token& peek() {
        if (!has_peek_) { peek_ = checked(scan()); has_peek_ = true; }
        return peek_;
    }
    Token next_raw() { return scan(); }  // the token stream as the reference's Tokenizer yields it (shm_pbrt_tokenize)
    std::string where(int line) const { return name_ + ":" + std::to_string(line); }

private:
    Token checked(Token t) {
        if (t.kind == Token::BAD_QUOTE) fail(where(t.line) + ": unterminated string");
        return t;
    }
    static bool is_space(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r'; }  // tokenizer.rs:92
    Token scan() {
        for (;;) {
            while (i_ < s_.size() && is_space(s_[i_])) { if (s_[i_] == '\n') ++line_; ++i_; }
            if (i_ < s_.size() && s_[i_] == '#') { while (i_ < s_.size() && s_[i_] != '\n') ++i_; continue; }
            break;
        }
        Token t;
        t.line = line_;
        if (i_ >= s_.size()) return t;
        const char c = s_[i_];
        if (c == '[') { ++i_; t.kind = Token::LBRACKET; return t; }
        if (c == ']') { ++i_; t.kind = Token::RBRACKET; return t; }
        if (c == '"') {  // tokenizer.rs:93-102: everything up to the next quote, line breaks included
            size_t j = i_ + 1;
            while (j < s_.size() && s_[j] != '"') { if (s_[j] == '\n') ++line_; ++j; }
            if (j >= s_.size()) { t.kind = Token::BAD_QUOTE; t.text = s_.substr(i_); i_ = s_.size(); return t; }
            t.kind = Token::STRING;
            t.text = s_.substr(i_ + 1, j - i_ - 1);
            i_ = j + 1;
            return t;
        }
        size_t j = i_;  // tokenizer.rs:108-116: a word ends at white space, a quote or a bracket ('#' only starts a comment between tokens)
        while (j < s_.size() && !is_space(s_[j]) && s_[j] != '[' && s_[j] != ']' && s_[j] != '"') ++j;
        t.kind = Token::WORD;
        t.text = s_.substr(i_, j - i_);
        i_ = j;
        return t;
    }
    const std::string& s_;
    std::string name_;
    size_t i_ = 0;
    int line_ = 1;
    bool has_peek_ = false;
    Token peek_;
};

// ---- parameter lists (loading/param.rs, paramdict.rs) -----------------------------------------------------------------------------
struct Param {
    std::string type, name;
    std::vector<float> f;
    std::vector<int> i;
    std::vector<std::string> s;
    std::vector<bool> b;
    int line = 0;
    int color_space = -1;  // ParsedParameter::color_space (paramdict.rs): the graphics state's colour space when the parameter was read
};
struct Params {
    std::vector<Param> v;
    const Param* find(const std::string& name) const {
        for (const Param& p : v) if (p.name == name) return &p;
        return nullptr;
    }
    float one_float(const std::string& n, float d) const { const Param* p = find(n); return (p && !p->f.empty()) ? p->f[0] : d; }
    int one_int(const std::string& n, int d) const { const Param* p = find(n); return (p && !p->i.empty()) ? p->i[0] : d; }
    bool one_bool(const std::string& n, bool d) const { const Param* p = find(n); return (p && !p->b.empty()) ? (bool)p->b[0] : d; }
    std::string one_string(const std::string& n, const std::string& d) const { const Param* p = find(n); return (p && !p->s.empty()) ? p->s[0] : d; }
    std::vector<float> floats(const std::string& n) const { const Param* p = find(n); return p ? p->f : std::vector<float>(); }
    std::vector<int> ints(const std::string& n) const { const Param* p = find(n); return p ? p->i : std::vector<int>(); }
};

static float parse_float(const Token& t, Tokenizer& tk) {
    char* end = nullptr;
    const float v = strtof(t.text.c_str(), &end);
    if (t.kind != Token::WORD || end == t.text.c_str() || *end) fail(tk.where(t.line) + ": expected a number, got \"" + t.text + "\"");
    return v;
}
static Params parse_params(Tokenizer& tk) {
    Params ps;
    while (tk.peek().kind == Token::STRING) {
        const Token decl = tk.next();
        Param p;
        p.line = decl.line;
        std::istringstream is(decl.text);
        if (!(is >> p.type >> p.name)) fail(tk.where(decl.line) + ": parameter declaration \"" + decl.text + "\" is not \"type name\"");
        static const std::set<std::string> kTypes = {"bool", "integer", "float", "point2", "vector2", "point3", "vector3", "normal", "normal3", "spectrum", "rgb",
                                                      "blackbody", "string", "texture", "point", "vector"};
        if (!kTypes.count(p.type)) fail(tk.where(decl.line) + ": unknown parameter type \"" + p.type + "\"");
        std::vector<Token> vals;
        if (tk.peek().kind == Token::LBRACKET) {
            tk.next();
            while (tk.peek().kind != Token::RBRACKET) {
                if (tk.peek().kind == Token::END) fail(tk.where(decl.line) + ": unterminated parameter list");
                // parser.rs:603-606: a directive before the closing bracket is an UnexpectedToken, not a value
                if (tk.peek().kind == Token::WORD && is_directive_name(tk.peek().text)) fail(tk.where(tk.peek().line) + ": unexpected directive " + tk.peek().text + " inside [ ]: missing closing bracket");
                vals.push_back(tk.next());
            }
            tk.next();
        } else {
            const Token v = tk.next();
            if (v.kind != Token::WORD && v.kind != Token::STRING) fail(tk.where(decl.line) + ": parameter \"" + p.name + "\" has no value");
            vals.push_back(v);
        }
        for (const Token& v : vals) {
            if (p.type == "integer") { char* e = nullptr; long x = strtol(v.text.c_str(), &e, 10); if (v.kind != Token::WORD || *e) fail(tk.where(v.line) + ": expected an integer"); p.i.push_back((int)x); }
            else if (p.type == "bool") { if (v.text != "true" && v.text != "false") fail(tk.where(v.line) + ": expected true or false"); p.b.push_back(v.text == "true"); }
            else if (p.type == "string" || p.type == "texture") { if (v.kind != Token::STRING) fail(tk.where(v.line) + ": expected a quoted string"); p.s.push_back(v.text); }
            else if (p.type == "spectrum" && v.kind == Token::STRING) p.s.push_back(v.text);
            else p.f.push_back(parse_float(v, tk));
        }
        if (ps.find(p.name)) fail(tk.where(decl.line) + ": duplicated parameter name \"" + p.name + "\"");  // param.rs:133-139 ParamList::add
        ps.v.push_back(p);
    }
    return ps;
}

// ---- graphics state (loading/scene.rs:1100-1220) -----------------------------------------------------------------------------------
struct GraphicsState {
    Xf ctm = xf_identity();
    bool reverse_orientation = false;
    int color_space = CS_SRGB;   // scene.rs:1119, 1561-1564: the ColorSpace directive's, part of the attribute state
    int material = 0;            // index into Assembly::materials
    Assembly::Emission area_light;
    Params shape_attributes, light_attributes, material_attributes, texture_attributes;  // Attribute "target" ... (scene.rs:1714-1730)
};
// ParameterDictionary::new_with_unowned (paramdict.rs:440-455): the directive's own parameters, last one first, then the target's attributes,
// last one first — a lookup takes the first match, so the directive overrides the attribute and a repeated name resolves to its last value
static Params with_attributes(const Params& own, const Params& attributes) {
    Params out;
    out.v.assign(own.v.rbegin(), own.v.rend());
    out.v.insert(out.v.end(), attributes.v.rbegin(), attributes.v.rend());
    return out;
}

class Loader {
public:
    Loader(const std::string& base_dir) : base_dir_(base_dir), a_(new Assembly()) {
        memset(&a_->camera, 0, sizeof(a_->camera));
        memset(&a_->film, 0, sizeof(a_->film));
        // BasicSceneBuilder::new (scene.rs:1221-1304): the default material is "diffuse" with default parameters
        gs_.material = make_material("diffuse", Params(), 0);
        // A colour space's coefficient table (rgb_to_spectra.rs:27-45 reads rgbtospec/<name>.spec from the working directory): an explicit
        // file first ($SHM_RGB2SPEC_<NAME>), then the reference's own location (beside the scene, then the working directory), then the table
        // tools/gen_rgb2spec.py leaves beside this library (Assembly::need_color_space)
        a_->scene_dir = base_dir;
        Dl_info info;
        if (dladdr(reinterpret_cast<const void*>(&shm_set_last_error), &info) && info.dli_fname) {
            std::string lib(info.dli_fname);
            const size_t slash = lib.find_last_of('/');
            a_->lib_data_dir = (slash == std::string::npos ? std::string(".") : lib.substr(0, slash)) + "/../data";
        }
        settings_.spp = 4;
        settings_.max_depth = 5;
    }
    struct Settings {
        int spp, max_depth;
        bool regularize = false, sample_lights = true, sample_bsdf = true;
        std::string integrator = "path", filename = "shimmer.pfm", sampler = "independent";
        int seed = 0;
        float white_balance = 0.0f;
        uint32_t render_space = SHM_RENDER_SPACE_CAMERA_WORLD;
        bool disable_pixel_jitter = false, disable_wavelength_jitter = false, force_diffuse = false, disable_texture_filtering = false;
    } settings_;

    // a directive's parameter list, each parameter stamped with the colour space in effect (ParameterDictionary::new(params,
    // graphics_state.color_space), scene.rs:1573-1654; Attribute stamps its own at declaration, scene.rs:1725-1729)
    Params read_params(Tokenizer& tk) {
        Params ps = parse_params(tk);
        for (Param& p : ps.v) p.color_space = gs_.color_space;
        return ps;
    }
    void parse(const std::string& text, const std::string& name) {
        Tokenizer tk(text, name);
        include_chain_.push_back(name);  // (an exception leaves the chain as it is: the loader object is discarded with it)
        for (;;) {
            const Token t = tk.next();
            if (t.kind == Token::END) break;
            if (t.kind != Token::WORD) fail(tk.where(t.line) + ": expected a directive");
            directive(t, tk);
        }
        include_chain_.pop_back();
    }
    int film_color_space() const { return film_color_space_; }
    std::unique_ptr<Assembly::Built> finish() {
        if (!world_) fail("the scene description has no WorldBegin");
        if (object_ != 0) fail("unmatched ObjectBegin" + (push_stack_.empty() ? std::string() : " from " + push_stack_.back().where));
        if (!stack_.empty()) fail("Missing end to AttributeBegin from " + push_stack_.back().where);  // scene.rs:2013-2017
        return Assembly::build(std::move(a_));
    }

private:
    std::string base_dir_;
    std::unique_ptr<Assembly> a_;
    GraphicsState gs_;
    std::vector<GraphicsState> stack_;
    // what pushed each entry of stack_ and where (scene.rs:1190-1192 `push_stack: Vec<(u8, FileLoc)>`, 'a' attribute / 'o' object):
    // AttributeEnd and ObjectEnd must find their own kind on top (scene.rs:1693-1712, 1929-1962)
    struct Pushed { char kind; std::string where; };
    std::vector<Pushed> push_stack_;
    std::vector<std::string> include_chain_;  // files being parsed right now (Include recursion: cycles and runaway depth are errors, not a native stack overflow)
    std::map<std::string, Xf> coordinate_systems_;
    std::map<std::string, int> named_materials_;
    std::map<std::string, uint32_t> float_texture_names_;
    // A named spectrum texture exists once per SpectrumType in the reference (scene.rs:268-294, 380-520: albedo / unbounded / illuminant
    // NamedTextures); here the albedo flavour is made at the Texture directive and the other two when a slot of that type first names it.
    // Definitions without RGB content are the same in the three flavours and share one node.
    struct SpectrumTexDef {
        std::string cls;
        Params ps;
        Xf rfo;
        int line = 0;
        bool type_dependent = false;
        bool have[3] = {false, false, false};
        ShmSpectrum variant[3];
    };
    std::map<std::string, SpectrumTexDef> spectrum_textures_;
    struct LoadedImage { uint32_t first_level, n_levels, n_channels, file_channels; bool color_space; };
    std::map<std::string, LoadedImage> image_cache_;  // TexInfo (texture.rs:77-84): filename + wrap + encoding -> the MIP pyramid's levels
    bool world_ = false;
    uint32_t object_ = 0;
    // pre-world entities (scene.rs:1578-1660): kept until WorldBegin, where the camera transform is known
    Params film_params_, camera_params_, filter_params_;
    std::string camera_type_ = "perspective", film_type_ = "rgb", filter_type_ = "box";
    int film_color_space_ = CS_SRGB;  // the colour space in effect at the Film directive (scene.rs:95): RgbFilm's output matrix uses its rgb_from_xyz
    Xf camera_from_world_ = xf_identity();
    Xf render_from_world_ = xf_identity();

    static void read_floats(Tokenizer& tk, float* out, int n) {
        const bool bracket = tk.peek().kind == Token::LBRACKET;
        if (bracket) tk.next();
        for (int k = 0; k < n; ++k) out[k] = parse_float(tk.next(), tk);
        if (bracket) { if (tk.next().kind != Token::RBRACKET) fail("expected ]"); }
    }
    static std::string read_string(Tokenizer& tk, const Token& at) {
        const Token t = tk.next();
        if (t.kind != Token::STRING) fail(tk.where(at.line) + ": " + at.text + " expects a quoted name");
        return t.text;
    }
    void need_world(const Token& t, Tokenizer& tk, bool want) const {
        if (world_ != want) fail(tk.where(t.line) + ": " + t.text + (want ? " is only allowed after WorldBegin" : " is only allowed before WorldBegin"));
    }
    Xf render_from_object() const { return xf_mul(render_from_world_, gs_.ctm); }  // scene.rs:1310-1313

    // ---- spectra from parameters (paramdict.rs:605-724) ----
    static SpectrumValue spectrum_of(const Param& p, Tokenizer& tk) {
        SpectrumValue v;
        std::ostringstream key;
        key << p.type;
        for (float x : p.f) key << ' ' << x;
        for (const std::string& x : p.s) key << ' ' << x;
        if (p.type == "rgb") key << " cs" << (p.color_space >= 0 ? p.color_space : 0);  // the same numbers in another colour space are another spectrum
        v.key = key.str();
        if (p.type == "float") { if (p.f.empty()) fail(tk.where(p.line) + ": empty float"); v.kind = SpectrumValue::CONSTANT; v.c = p.f[0]; }
        else if (p.type == "rgb") { if (p.f.size() != 3) fail(tk.where(p.line) + ": rgb needs three values"); v.kind = SpectrumValue::RGB; memcpy(v.rgb, p.f.data(), 12); v.color_space = p.color_space >= 0 ? p.color_space : CS_SRGB; }
        else if (p.type == "blackbody") { if (p.f.empty()) fail(tk.where(p.line) + ": blackbody needs a temperature"); v.kind = SpectrumValue::DENSE; v.dense = blackbody_dense(p.f[0]); }
        else if (p.type == "spectrum" && !p.f.empty()) {
            if (p.f.size() % 2) fail(tk.where(p.line) + ": Found odd number of values for " + p.name);
            v.kind = SpectrumValue::PIECEWISE;
            for (size_t i = 0; i + 1 < p.f.size(); i += 2) {
                if (i > 0 && p.f[i] <= v.lam.back()) fail(tk.where(p.line) + ": Spectrum description invalid: wavelengths aren't increasing");
                v.lam.push_back(p.f[i]);
                v.val.push_back(p.f[i + 1]);
            }
            if (v.lam.size() == 1) { v.kind = SpectrumValue::CONSTANT; v.c = 0.0f; fail(tk.where(p.line) + ": a spectrum needs at least two samples"); }
        } else if (p.type == "spectrum" && !p.s.empty()) {
            static const std::map<std::string, std::pair<const uint32_t*, size_t>> named = {
                {"glass-BK7", {TBL_GLASS_BK7_ETA_SAMPLES, sizeof(TBL_GLASS_BK7_ETA_SAMPLES) / 4}}, {"glass-baf10", {TBL_GLASS_BAF10_ETA_SAMPLES, sizeof(TBL_GLASS_BAF10_ETA_SAMPLES) / 4}},
                {"glass-F11", {TBL_GLASS_F11_ETA_SAMPLES, sizeof(TBL_GLASS_F11_ETA_SAMPLES) / 4}}, {"metal-Cu-eta", {TBL_CU_ETA_SAMPLES, sizeof(TBL_CU_ETA_SAMPLES) / 4}},
                {"metal-Cu-k", {TBL_CU_K_SAMPLES, sizeof(TBL_CU_K_SAMPLES) / 4}}, {"metal-Au-eta", {TBL_AU_ETA_SAMPLES, sizeof(TBL_AU_ETA_SAMPLES) / 4}},
                {"metal-Au-k", {TBL_AU_K_SAMPLES, sizeof(TBL_AU_K_SAMPLES) / 4}}, {"metal-Ag-eta", {TBL_AG_ETA_SAMPLES, sizeof(TBL_AG_ETA_SAMPLES) / 4}},
                {"metal-Ag-k", {TBL_AG_K_SAMPLES, sizeof(TBL_AG_K_SAMPLES) / 4}}, {"metal-Al-eta", {TBL_AL_ETA_SAMPLES, sizeof(TBL_AL_ETA_SAMPLES) / 4}},
                {"metal-Al-k", {TBL_AL_K_SAMPLES, sizeof(TBL_AL_K_SAMPLES) / 4}}};
            if (p.s[0] == "StdIllum-D65") { v.kind = SpectrumValue::DENSE; v.dense = illuminant_d65_dense(); return v; }
            auto it = named.find(p.s[0]);
            v.kind = SpectrumValue::PIECEWISE;
            if (it == named.end()) {  // Spectrum::read_from_file -> PiecewiseLinearSpectrum::read (spectrum.rs:90-108, 372-398): wavelength value pairs
                std::ifstream in(p.s[0]);
                if (!in) fail(tk.where(p.line) + ": Unable to read/invalid spectrum file " + p.s[0]);
                std::vector<float> vals;
                std::string word;
                while (in >> word) {
                    char* end = nullptr;
                    vals.push_back(strtof(word.c_str(), &end));
                    if (end == word.c_str() || *end) fail(tk.where(p.line) + ": " + p.s[0] + ": Unable to parse float value!");
                }
                if (vals.empty() || vals.size() % 2) fail(tk.where(p.line) + ": Unable to read/invalid spectrum file " + p.s[0]);
                for (size_t i = 0; i + 1 < vals.size(); i += 2) {
                    if (i > 0 && vals[i] <= v.lam.back()) fail(tk.where(p.line) + ": " + p.s[0] + ": Spectrum file invalid, wavelengths not increasing");
                    v.lam.push_back(vals[i]);
                    v.val.push_back(vals[i + 1]);
                }
                if (v.lam.size() < 2) fail(tk.where(p.line) + ": " + p.s[0] + ": a spectrum needs at least two samples");
                return v;
            }
            from_interleaved(table(it->second.first, it->second.second), v.lam, v.val);
        } else if (p.type == "texture") { if (p.s.empty()) fail(tk.where(p.line) + ": texture parameter without a name"); v.kind = SpectrumValue::TEXTURE; v.texture = p.s[0]; }
        else fail(tk.where(p.line) + ": parameter \"" + p.name + "\" is not a spectrum");
        return v;
    }
    static SpectrumValue named_value(const char* name) {  // the default metals (material.rs:385-397)
        Param p;
        p.type = "spectrum";
        p.name = "eta";
        p.s.push_back(name);
        Tokenizer dummy(p.name, "default");
        return spectrum_of(p, dummy);
    }
    ShmSpectrum slot(const Params& ps, const std::string& name, Tokenizer* tk, const SpectrumValue* dflt, SpectrumType type) {
        static const std::map<std::string, ShmSpectrum> none;
        const Param* p = ps.find(name);
        if (!p) return a_->bind(*dflt, none, type);
        Param q = *p;
        if (q.type == "spectrum" && !q.s.empty() && q.f.empty() && q.s[0].find('.') != std::string::npos) q.s[0] = resolve(q.s[0]);  // a spectrum file
        const SpectrumValue v = spectrum_of(q, *tk);
        if (v.kind == SpectrumValue::TEXTURE) return spectrum_texture_ref(v.texture, type, *tk, p->line);
        try { return a_->bind(v, none, type); }
        catch (const LoadError& e) { fail(tk->where(p->line) + ": parameter \"" + name + "\": " + e.what(), e.code); }
    }
    // a float parameter or the float texture bound to it: sets ShmMaterial::float_tex[slot] when it names a texture
    float float_or_texture(const Params& ps, const std::string& name, float dflt, ShmMaterial& m, int slot_index, Tokenizer* tk) {
        const Param* p = ps.find(name);
        if (!p) return dflt;
        if (p->type == "texture") {
            auto it = float_texture_names_.find(p->s.empty() ? "" : p->s[0]);
            if (it == float_texture_names_.end()) fail(tk->where(p->line) + ": Couldn't find float texture named \"" + (p->s.empty() ? "" : p->s[0]) + "\"");
            m.float_tex[slot_index] = it->second + 1u;
            return dflt;
        }
        if (p->f.empty()) fail(tk->where(p->line) + ": parameter \"" + name + "\" needs a value");
        return p->f[0];
    }

    // ---- Material::create (material.rs:56-110 + the per-material create functions) ----
    int make_material(const std::string& type, const Params& ps, Tokenizer* tk) {
        Tokenizer dummy(type, "default");
        if (!tk) tk = &dummy;
        ShmMaterial m;
        memset(&m, 0, sizeof(m));
        SpectrumValue c05, c0, c15;
        c05.kind = c0.kind = c15.kind = SpectrumValue::CONSTANT;
        c05.c = 0.5f; c0.c = 0.0f; c15.c = 1.5f;
        const ShmSpectrum zero = Assembly::spec_constant(0.0f);
        m.a = m.b = m.c = m.d = zero;
        auto roughness = [&](const std::string& prefix, float& u, float& v, int su, int sv) {  // material.rs:400-412
            const float r = float_or_texture(ps, prefix + "roughness", 0.0f, m, su, tk);
            const bool r_tex = m.float_tex[su] != 0;
            const uint32_t r_handle = m.float_tex[su];
            m.float_tex[su] = 0;
            u = float_or_texture(ps, prefix + "uroughness", r, m, su, tk);
            if (!ps.find(prefix + "uroughness") && r_tex) m.float_tex[su] = r_handle;
            v = float_or_texture(ps, prefix + "vroughness", r, m, sv, tk);
            if (!ps.find(prefix + "vroughness") && r_tex) m.float_tex[sv] = r_handle;
        };
        auto displacement = [&](bool always) {  // material.rs:280 (Diffuse installs a constant 0) vs get_float_texture_or_none elsewhere
            if (always || ps.find("displacement")) {
                m.has_displacement = 1;
                m.displacement = float_or_texture(ps, "displacement", 0.0f, m, SHM_FLOATSLOT_DISPLACEMENT, tk);
            }
        };
        const std::string normal_map_file = resolve(ps.one_string("normalmap", ""));
        if (type == "diffuse") {
            m.kind = SHM_MATERIAL_DIFFUSE;
            m.a = slot(ps, "reflectance", tk, &c05, SPECTRUM_ALBEDO);
            displacement(true);
        } else if (type == "conductor") {
            m.kind = SHM_MATERIAL_CONDUCTOR;
            if (ps.find("reflectance")) fail("conductor \"reflectance\" is not representable in ShmMaterial: give eta and k", SHM_ERR_UNSUPPORTED);
            const SpectrumValue cu_eta = named_value("metal-Cu-eta"), cu_k = named_value("metal-Cu-k");
            m.a = slot(ps, "eta", tk, &cu_eta, SPECTRUM_UNBOUNDED);
            m.b = slot(ps, "k", tk, &cu_k, SPECTRUM_UNBOUNDED);
            roughness("", m.u_roughness, m.v_roughness, SHM_FLOATSLOT_U_ROUGHNESS, SHM_FLOATSLOT_V_ROUGHNESS);
            m.remap_roughness = ps.one_bool("remaproughness", true);
            displacement(false);
        } else if (type == "dielectric" || type == "thindielectric") {
            m.kind = type == "dielectric" ? SHM_MATERIAL_DIELECTRIC : SHM_MATERIAL_THIN_DIELECTRIC;
            m.a = slot(ps, "eta", tk, &c15, SPECTRUM_UNBOUNDED);  // material.rs:546-553: a float "eta" is a constant spectrum, default 1.5
            if (m.a.kind >= SHM_SPECTRUM_IMAGE_TEXTURE) fail("dielectric eta cannot be a texture");
            if (type == "dielectric") {
                roughness("", m.u_roughness, m.v_roughness, SHM_FLOATSLOT_U_ROUGHNESS, SHM_FLOATSLOT_V_ROUGHNESS);
                m.remap_roughness = ps.one_bool("remaproughness", true);
            }
            displacement(false);
        } else if (type == "coateddiffuse") {
            m.kind = SHM_MATERIAL_COATED_DIFFUSE;
            m.a = slot(ps, "reflectance", tk, &c05, SPECTRUM_ALBEDO);
            roughness("", m.u_roughness, m.v_roughness, SHM_FLOATSLOT_U_ROUGHNESS, SHM_FLOATSLOT_V_ROUGHNESS);
            m.thickness = float_or_texture(ps, "thickness", 0.01f, m, SHM_FLOATSLOT_THICKNESS, tk);
            m.d = slot(ps, "eta", tk, &c15, SPECTRUM_UNBOUNDED);
            m.max_depth = ps.one_int("maxdepth", 10);
            m.n_samples = ps.one_int("nsamples", 1);
            m.g = float_or_texture(ps, "g", 0.0f, m, SHM_FLOATSLOT_G, tk);
            m.c = slot(ps, "albedo", tk, &c0, SPECTRUM_ALBEDO);
            m.remap_roughness = ps.one_bool("remaproughness", true);
            displacement(false);
        } else if (type == "coatedconductor") {
            m.kind = SHM_MATERIAL_COATED_CONDUCTOR;
            roughness("interface.", m.u_roughness, m.v_roughness, SHM_FLOATSLOT_U_ROUGHNESS, SHM_FLOATSLOT_V_ROUGHNESS);
            roughness("conductor.", m.u2_roughness, m.v2_roughness, SHM_FLOATSLOT_U2_ROUGHNESS, SHM_FLOATSLOT_V2_ROUGHNESS);
            m.thickness = float_or_texture(ps, "thickness", 0.01f, m, SHM_FLOATSLOT_THICKNESS, tk);
            m.d = slot(ps, "interface.eta", tk, &c15, SPECTRUM_UNBOUNDED);
            if (ps.find("reflectance")) {
                if (ps.find("conductor.eta") || ps.find("k")) fail("Cannot specify both reflectance and conductor eta/k for conductor material.");
                m.conductor_from_reflectance = 1;
                m.a = slot(ps, "reflectance", tk, &c05, SPECTRUM_ALBEDO);
            } else {
                const SpectrumValue cu_eta = named_value("metal-Cu-eta"), cu_k = named_value("metal-Cu-k");
                m.a = slot(ps, "conductor.eta", tk, &cu_eta, SPECTRUM_UNBOUNDED);
                m.b = slot(ps, "k", tk, &cu_k, SPECTRUM_UNBOUNDED);
            }
            m.max_depth = ps.one_int("maxdepth", 10);
            m.n_samples = ps.one_int("nsamples", 1);
            m.g = float_or_texture(ps, "g", 0.0f, m, SHM_FLOATSLOT_G, tk);
            m.c = slot(ps, "albedo", tk, &c0, SPECTRUM_ALBEDO);
            m.remap_roughness = ps.one_bool("remaproughness", true);
            displacement(false);
        } else if (type == "mix") {  // material.rs:56-110: "materials" names two NAMED materials
            m.kind = SHM_MATERIAL_MIX;
            const Param* names = ps.find("materials");
            if (!names || names->s.size() != 2) fail("Must provide two values for \"materials\" for mix material.");
            for (int k = 0; k < 2; ++k) {
                auto it = named_materials_.find(names->s[k]);
                if (it == named_materials_.end()) fail(names->s[k] + ": named material not found.");
                m.mix_material[k] = (uint32_t)it->second;
            }
            m.mix_amount = float_or_texture(ps, "amount", 0.5f, m, SHM_FLOATSLOT_MIX_AMOUNT, tk);
        } else if (type == "interface" || type == "") {
            fail("\"interface\" materials (media boundaries) are todo!() in the reference and not supported", SHM_ERR_UNSUPPORTED);
        } else {
            fail("Material \"" + type + "\" unknown.");
        }
        if (!normal_map_file.empty() && type != "mix") {
            // load_normal_map (scene.rs:347-378): read with the LINEAR encoding, must have R, G, B; normal_map() reads only the finest level
            // with WrapMode::Repeat and bilerp (material.rs:1453-1475) — the ABI takes it as an image texture with those settings
            const LoadedImage im = load_image(normal_map_file, WRAP_REPEAT, "linear", *tk, 0, true);
            if (im.n_channels != 3) fail("Normal map \"" + normal_map_file + "\" should have RGB channels.");
            ShmImageTexture t = default_image_texture();
            t.n_channels = 3;
            t.first_level = im.first_level;
            t.n_levels = im.n_levels;
            a_->image_textures.push_back(t);
            m.normal_map = (uint32_t)a_->image_textures.size();
        }
        a_->materials.push_back(m);
        return (int)a_->materials.size() - 1;
    }

    // ---- Texture "name" "float|spectrum" "class" (scene.rs:1732-1805, texture.rs create functions) ----
    uint32_t float_texture_operand(const Params& ps, const std::string& name, float dflt, Tokenizer& tk) {
        const Param* p = ps.find(name);
        if (p && p->type == "texture") {
            auto it = float_texture_names_.find(p->s.empty() ? "" : p->s[0]);
            if (it == float_texture_names_.end()) fail(tk.where(p->line) + ": Couldn't find float texture named \"" + (p->s.empty() ? "" : p->s[0]) + "\"");
            return it->second;
        }
        ShmFloatTexture t;
        memset(&t, 0, sizeof(t));
        t.kind = SHM_FLOATTEX_CONSTANT;
        t.value = (p && !p->f.empty()) ? p->f[0] : dflt;
        a_->float_textures.push_back(t);
        return (uint32_t)a_->float_textures.size() - 1;
    }
    uint32_t spectrum_texture_operand(const Params& ps, const std::string& name, float dflt, Tokenizer& tk, SpectrumType type) {
        SpectrumValue d;
        d.kind = SpectrumValue::CONSTANT;
        d.c = dflt;
        const ShmSpectrum sp = slot(ps, name, &tk, &d, type);
        if (sp.kind == SHM_SPECTRUM_TEXTURE_NODE) return sp.offset;
        ShmSpectrumTexture t;
        memset(&t, 0, sizeof(t));
        t.kind = SHM_SPECTEX_LEAF;
        t.leaf = sp;
        a_->spectrum_textures.push_back(t);
        return (uint32_t)a_->spectrum_textures.size() - 1;
    }
    // ---- images: Image::read -> MIPMap::create_from_file -> Image::generate_pyramid (host/image_io.hpp) -> the ABI's level / texel tables ----
    static ShmImageTexture default_image_texture() {
        ShmImageTexture t;
        memset(&t, 0, sizeof(t));
        t.mapping = SHM_TEXMAP_UV;
        t.su = t.sv = 1.0f;
        t.vs[0] = 1.0f; t.vt[1] = 1.0f;
        const M4 id = m4_identity();
        memcpy(t.texture_from_render, id.m, sizeof(float) * 16);
        t.filter = SHM_TEXFILTER_BILINEAR;
        t.max_anisotropy = 8.0f;
        t.wrap = SHM_WRAP_REPEAT;
        t.scale = 1.0f;
        t.n_channels = 3;
        return t;
    }
    LoadedImage load_image(const std::string& file, WrapMode wrap, const std::string& encoding, Tokenizer& tk, int line, bool finest_level_only) {
        const std::string key = file + "|" + std::to_string((int)wrap) + "|" + encoding + (finest_level_only ? "|0" : "");
        auto it = image_cache_.find(key);
        if (it != image_cache_.end()) return it->second;
        LoadedImage out;
        try {
            HostImage image = image_read(file, ColorEncoding::get(encoding));
            std::vector<HostImage> pyramid;
            if (finest_level_only) pyramid.push_back(image);
            else pyramid = generate_pyramid(mipmap_select_channels(image), wrap);
            out.first_level = (uint32_t)a_->image_levels.size();
            out.n_levels = (uint32_t)pyramid.size();
            out.color_space = image.srgb_color_space;
            // texel_rgb / bilerp read channels 0..2 of a 4-channel image (mipmap.rs:203-231, 317-331): the alpha plane is not handed over
            const int keep = pyramid[0].nc == 4 ? 3 : pyramid[0].nc;
            out.n_channels = (uint32_t)keep;
            for (const HostImage& lv : pyramid) {
                ShmImageLevel l;
                memset(&l, 0, sizeof(l));
                l.width = lv.res[0];
                l.height = lv.res[1];
                if (a_->texels.size() > 0xffffffffull) fail("image textures: more than 2^32 texel floats");
                l.texel_offset = (uint32_t)a_->texels.size();
                a_->image_levels.push_back(l);
                for (int y = 0; y < lv.res[1]; ++y) for (int x = 0; x < lv.res[0]; ++x) for (int c = 0; c < keep; ++c) a_->texels.push_back(lv.get(x, y, c));
            }
            out.file_channels = (uint32_t)pyramid[0].nc;  // (4: an alpha plane that is not all ones — FloatImageTexture's bilerp would read it)
        } catch (const LoadError& e) { fail(tk.where(line) + ": " + e.what(), e.code); }
        image_cache_[key] = out;
        return out;
    }
    // ImageTextureBase parameters (texture.rs:345-391, 728-775) + TextureMapping2D::create (texture.rs:846-880)
    uint32_t image_texture(const Params& ps, const Xf& render_from_texture, SpectrumType type, bool spectrum, Tokenizer& tk, int line) {
        ShmImageTexture t = default_image_texture();
        const std::string mapping = ps.one_string("mapping", "uv");
        if (mapping == "uv") {
            t.mapping = SHM_TEXMAP_UV;
            t.su = ps.one_float("uscale", 1.0f); t.sv = ps.one_float("vscale", 1.0f);
            t.du = ps.one_float("udelta", 0.0f); t.dv = ps.one_float("vdelta", 0.0f);
        } else if (mapping == "spherical" || mapping == "cylindrical" || mapping == "planar") {
            t.mapping = mapping == "spherical" ? SHM_TEXMAP_SPHERICAL : (mapping == "cylindrical" ? SHM_TEXMAP_CYLINDRICAL : SHM_TEXMAP_PLANAR);
            memcpy(t.texture_from_render, render_from_texture.inv.m, sizeof(float) * 16);  // render_from_texture.inverse()
            if (mapping == "planar") {
                const std::vector<float> v1 = ps.floats("v1"), v2 = ps.floats("v2");
                if (v1.size() == 3) memcpy(t.vs, v1.data(), 12);
                if (v2.size() == 3) memcpy(t.vt, v2.data(), 12);
                t.du = ps.one_float("udelta", 0.0f); t.dv = ps.one_float("vdelta", 0.0f);
            }
        } else fail(tk.where(line) + ": Unknown texture mapping type " + mapping);
        t.max_anisotropy = ps.one_float("maxanisotropy", 8.0f);
        const std::string filter = ps.one_string("filter", "bilinear");  // FilterFunction::parse (mipmap.rs:343-360)
        if (filter == "point") t.filter = SHM_TEXFILTER_POINT;
        else if (filter == "bilinear") t.filter = SHM_TEXFILTER_BILINEAR;
        else if (filter == "trilinear") t.filter = SHM_TEXFILTER_TRILINEAR;
        else if (filter == "ewa" || filter == "EWA") t.filter = SHM_TEXFILTER_EWA;
        else fail(tk.where(line) + ": Unknown filter function " + filter);
        const std::string wrap = ps.one_string("wrap", "repeat");  // WrapMode::parse (image.rs:81-94)
        if (wrap == "repeat") t.wrap = SHM_WRAP_REPEAT;
        else if (wrap == "clamp") t.wrap = SHM_WRAP_CLAMP;
        else if (wrap == "black") t.wrap = SHM_WRAP_BLACK;
        else if (wrap == "octahedralsphere") t.wrap = SHM_WRAP_OCTAHEDRAL_SPHERE;
        else fail(tk.where(line) + ": Unknown wrap mode " + wrap);
        t.scale = ps.one_float("scale", 1.0f);
        t.invert = ps.one_bool("invert", false) ? 1 : 0;
        const std::string file = resolve(ps.one_string("filename", ""));
        if (file.empty()) fail(tk.where(line) + ": No filename provided for texture.");
        const size_t dot = file.find_last_of('.');
        const std::string encoding = ps.one_string("encoding", (dot != std::string::npos && file.substr(dot + 1) == "png") ? "sRGB" : "linear");
        const LoadedImage im = load_image(file, (WrapMode)t.wrap, encoding, tk, line, false);
        t.first_level = im.first_level;
        t.n_levels = im.n_levels;
        t.n_channels = (uint8_t)im.n_channels;
        t.spectrum_type = (uint8_t)type;
        if (spectrum) {
            t.has_color_space = im.color_space ? 1 : 0;
            if (im.color_space) a_->need_color_space();
        } else if (im.file_channels == 4) {
            // TexelType for Float reads channel 0 in texel() but the ALPHA channel in bilerp() of a 4-channel image (mipmap.rs:296-309)
            fail(tk.where(line) + ": float imagemap of an RGBA image with a non-opaque alpha channel is not supported", SHM_ERR_UNSUPPORTED);
        }
        a_->image_textures.push_back(t);
        return (uint32_t)a_->image_textures.size() - 1;
    }
    static bool params_have_rgb(const Params& ps) {
        for (const Param& p : ps.v) if (p.type == "rgb") return true;
        return false;
    }
    // one flavour of a named spectrum texture (SpectrumTexture::create, texture.rs:420-503)
    ShmSpectrum make_spectrum_texture(const SpectrumTexDef& def, SpectrumType type, Tokenizer& tk) {
        const Params& ps = def.ps;
        const std::string& cls = def.cls;
        const int line = def.line;
        ShmSpectrumTexture t;
        memset(&t, 0, sizeof(t));
        if (cls == "constant") {
            SpectrumValue one;
            one.kind = SpectrumValue::CONSTANT;
            one.c = 1.0f;
            return slot(ps, "value", &tk, &one, type);
        }
        if (cls == "imagemap") {
            ShmSpectrum sp;
            memset(&sp, 0, sizeof(sp));
            sp.kind = SHM_SPECTRUM_IMAGE_TEXTURE;
            sp.offset = image_texture(ps, def.rfo, type, true, tk, line);
            return sp;
        }
        if (cls == "scale") { t.kind = SHM_SPECTEX_SCALED; t.a = spectrum_texture_operand(ps, "tex", 1.0f, tk, type); t.f = float_texture_operand(ps, "scale", 1.0f, tk); }
        else if (cls == "mix") { t.kind = SHM_SPECTEX_MIX; t.a = spectrum_texture_operand(ps, "tex1", 0.0f, tk, type); t.b = spectrum_texture_operand(ps, "tex2", 1.0f, tk, type); t.f = float_texture_operand(ps, "amount", 0.5f, tk); }
        else if (cls == "directionmix") {
            t.kind = SHM_SPECTEX_DIRECTION_MIX;
            t.a = spectrum_texture_operand(ps, "tex1", 0.0f, tk, type);
            t.b = spectrum_texture_operand(ps, "tex2", 1.0f, tk, type);
            const std::vector<float> d = ps.floats("dir");
            const V3 dir = xf_vector(def.rfo.m, d.size() == 3 ? shm::v3(d[0], d[1], d[2]) : shm::v3(0.0f, 1.0f, 0.0f));
            t.dir[0] = dir.x; t.dir[1] = dir.y; t.dir[2] = dir.z;
        } else if (cls == "ptex") fail(tk.where(line) + ": ptex textures are not part of the reference", SHM_ERR_UNSUPPORTED);
        else fail(tk.where(line) + ": Texture " + cls + " unknown");
        a_->spectrum_textures.push_back(t);
        ShmSpectrum sp;
        memset(&sp, 0, sizeof(sp));
        sp.kind = SHM_SPECTRUM_TEXTURE_NODE;
        sp.offset = (uint32_t)a_->spectrum_textures.size() - 1;
        return sp;
    }
    ShmSpectrum spectrum_texture_ref(const std::string& name, SpectrumType type, Tokenizer& tk, int line) {
        auto it = spectrum_textures_.find(name);
        if (it == spectrum_textures_.end()) fail(tk.where(line) + ": Couldn't find spectrum texture named \"" + name + "\"");
        SpectrumTexDef& def = it->second;
        if (!def.have[type]) {
            def.variant[type] = make_spectrum_texture(def, type, tk);
            def.have[type] = true;
        }
        return def.variant[type];
    }
    void texture(const std::string& name, const std::string& ty, const std::string& cls, const Params& ps, Tokenizer& tk, int line) {
        if (ty == "float") {
            if (float_texture_names_.count(name)) fail(tk.where(line) + ": Texture \"" + name + "\" being redefined");
            ShmFloatTexture t;
            memset(&t, 0, sizeof(t));
            if (cls == "constant") { t.kind = SHM_FLOATTEX_CONSTANT; t.value = ps.one_float("value", 1.0f); }
            else if (cls == "scale") { t.kind = SHM_FLOATTEX_SCALED; t.a = float_texture_operand(ps, "tex", 1.0f, tk); t.b = float_texture_operand(ps, "scale", 1.0f, tk); }
            else if (cls == "mix") { t.kind = SHM_FLOATTEX_MIX; t.a = float_texture_operand(ps, "tex1", 0.0f, tk); t.b = float_texture_operand(ps, "tex2", 1.0f, tk); t.c = float_texture_operand(ps, "amount", 0.5f, tk); }
            else if (cls == "directionmix") {
                t.kind = SHM_FLOATTEX_DIRECTION_MIX;
                t.a = float_texture_operand(ps, "tex1", 0.0f, tk);
                t.b = float_texture_operand(ps, "tex2", 1.0f, tk);
                const std::vector<float> d = ps.floats("dir");
                const V3 dir = xf_vector(render_from_object().m, d.size() == 3 ? shm::v3(d[0], d[1], d[2]) : shm::v3(0.0f, 1.0f, 0.0f));  // texture.rs:265-270: in render space
                t.dir[0] = dir.x; t.dir[1] = dir.y; t.dir[2] = dir.z;
            } else if (cls == "imagemap") { t.kind = SHM_FLOATTEX_IMAGE; t.image = image_texture(ps, render_from_object(), SPECTRUM_ALBEDO, false, tk, line); }
            else if (cls == "ptex") fail(tk.where(line) + ": ptex textures are not part of the reference", SHM_ERR_UNSUPPORTED);
            else fail(tk.where(line) + ": Texture " + cls + " unknown");
            a_->float_textures.push_back(t);
            float_texture_names_[name] = (uint32_t)a_->float_textures.size() - 1;
        } else if (ty == "spectrum") {
            if (spectrum_textures_.count(name)) fail(tk.where(line) + ": Texture \"" + name + "\" being redefined");
            SpectrumTexDef def;
            def.cls = cls;
            def.ps = ps;
            def.rfo = render_from_object();
            def.line = line;
            def.type_dependent = cls == "imagemap" || params_have_rgb(ps);
            for (const Param& p : ps.v)
                if (p.type == "texture" && !p.s.empty()) { auto it = spectrum_textures_.find(p.s[0]); if (it != spectrum_textures_.end() && it->second.type_dependent) def.type_dependent = true; }
            def.variant[SPECTRUM_ALBEDO] = make_spectrum_texture(def, SPECTRUM_ALBEDO, tk);
            def.have[SPECTRUM_ALBEDO] = true;
            if (!def.type_dependent) { def.variant[SPECTRUM_UNBOUNDED] = def.variant[SPECTRUM_ILLUMINANT] = def.variant[SPECTRUM_ALBEDO]; def.have[SPECTRUM_UNBOUNDED] = def.have[SPECTRUM_ILLUMINANT] = true; }
            spectrum_textures_[name] = def;
        } else fail(tk.where(line) + ": texture type \"" + ty + "\" unknown (float or spectrum)");
    }

    // ---- Shape (scene.rs:1321-1373 + shape/*.rs create functions) ----
    void shape(const std::string& type, const Params& ps, Tokenizer& tk, int line) {
        const Xf rfo = render_from_object();
        const bool reverse = gs_.reverse_orientation;
        auto transformed_mesh = [&](const std::vector<float>& P, const std::vector<float>& N, const std::vector<float>& S, const std::vector<float>& UV) {
            Assembly::Mesh m;  // TriangleMesh::new / BilinearPatchMesh::new (mesh.rs:22-70, 305-350): everything into render space
            m.p.resize(P.size());
            for (size_t i = 0; i + 2 < P.size(); i += 3) { const V3 q = xf_point(rfo.m, shm::v3(P[i], P[i + 1], P[i + 2])); m.p[i] = q.x; m.p[i + 1] = q.y; m.p[i + 2] = q.z; }
            m.n.resize(N.size());
            for (size_t i = 0; i + 2 < N.size(); i += 3) {
                V3 q = xf_normal(rfo.inv, shm::v3(N[i], N[i + 1], N[i + 2]));
                if (reverse) q = -q;
                m.n[i] = q.x; m.n[i + 1] = q.y; m.n[i + 2] = q.z;
            }
            m.s.resize(S.size());
            for (size_t i = 0; i + 2 < S.size(); i += 3) { const V3 q = xf_vector(rfo.m, shm::v3(S[i], S[i + 1], S[i + 2])); m.s[i] = q.x; m.s[i + 1] = q.y; m.s[i + 2] = q.z; }
            m.uv = UV;
            m.reverse = reverse;
            m.swaps = swaps_handedness(rfo.m);
            return m;
        };
        if (type == "sphere") {
            const float radius = ps.one_float("radius", 1.0f);
            a_->add_sphere(radius, ps.one_float("zmin", -radius), ps.one_float("zmax", radius), ps.one_float("phimax", 360.0f), rfo, reverse, (uint32_t)gs_.material,
                           gs_.area_light, object_);
        } else if (type == "trianglemesh") {  // triangle.rs:55-140
            const std::vector<float> P = ps.floats("P");
            std::vector<int> vi = ps.ints("indices");
            if (vi.empty()) {
                if (P.size() == 9) vi = {0, 1, 2};
                else fail(tk.where(line) + ": Vertex indices \"indices\" must be provided with a triangle mesh.");
            }
            if (vi.size() % 3) fail(tk.where(line) + ": Number of vertex indices not a multiple of 3");
            if (P.empty() || P.size() % 3) fail(tk.where(line) + ": Vertex positions \"P\" must be provided with a triangle mesh.");
            const size_t nv = P.size() / 3;
            const std::vector<float> UV = ps.floats("uv"), S = ps.floats("S"), N = ps.floats("N");
            if (!UV.empty() && UV.size() != 2 * nv) fail(tk.where(line) + ": Number of \"uv\"s for triangle mesh must match \"P\"s.");
            if (!S.empty() && S.size() != 3 * nv) fail(tk.where(line) + ": Number of \"S\"s for triangle mesh must match \"P\"s.");
            if (!N.empty() && N.size() != 3 * nv) fail(tk.where(line) + ": Number of \"N\"s for triangle mesh must match \"P\"s.");
            Assembly::Mesh m = transformed_mesh(P, N, S, UV);
            for (int v : vi) { if (v < 0 || (size_t)v >= nv) fail(tk.where(line) + ": trianglemesh has out-of-bounds vertex index"); m.vi.push_back((uint32_t)v); }
            a_->add_mesh(std::move(m), (uint32_t)gs_.material, gs_.area_light, object_);
        } else if (type == "bilinearmesh") {  // bilinear_patch.rs create
            const std::vector<float> P = ps.floats("P");
            std::vector<int> vi = ps.ints("indices");
            if (vi.empty()) { if (P.size() == 12) vi = {0, 1, 2, 3}; else fail(tk.where(line) + ": Vertex indices \"indices\" must be provided with a bilinear patch mesh."); }
            if (vi.size() % 4) fail(tk.where(line) + ": Number of vertex indices not a multiple of 4");
            if (P.empty() || P.size() % 3) fail(tk.where(line) + ": Vertex positions \"P\" must be provided with a bilinear patch mesh.");
            const size_t nv = P.size() / 3;
            const std::vector<float> UV = ps.floats("uv"), N = ps.floats("N");
            if (!UV.empty() && UV.size() != 2 * nv) fail(tk.where(line) + ": Number of \"uv\"s must match \"P\"s.");
            if (!N.empty() && N.size() != 3 * nv) fail(tk.where(line) + ": Number of \"N\"s must match \"P\"s.");
            Assembly::Mesh m = transformed_mesh(P, N, std::vector<float>(), UV);
            for (int v : vi) { if (v < 0 || (size_t)v >= nv) fail(tk.where(line) + ": bilinearmesh has out-of-bounds vertex index"); m.vi.push_back((uint32_t)v); }
            a_->add_patch_mesh(std::move(m), (uint32_t)gs_.material, gs_.area_light, object_);
        } else if (type == "plymesh") {  // shape/shape.rs:97-135
            const std::string fn = resolve(ps.one_string("filename", ""));
            ShmPlyMesh pm;
            if (shm_ply_read(fn.c_str(), &pm) != SHM_OK) fail(tk.where(line) + ": " + shm_last_error());
            const std::vector<float> P(pm.p, pm.p + 3 * pm.n_vertices), N(pm.n, pm.n + 3 * pm.n_vertices), UV(pm.uv, pm.uv + 2 * pm.n_vertices);
            const std::vector<int32_t> tri(pm.tri_indices, pm.tri_indices + pm.n_tri_indices), quad(pm.quad_indices, pm.quad_indices + pm.n_quad_indices);
            shm_ply_free(&pm);
            if (!tri.empty()) {
                Assembly::Mesh m = transformed_mesh(P, N, std::vector<float>(), UV);
                for (int32_t v : tri) m.vi.push_back((uint32_t)v);
                a_->add_mesh(std::move(m), (uint32_t)gs_.material, gs_.area_light, object_);
            }
            if (!quad.empty()) {
                Assembly::Mesh m = transformed_mesh(P, N, std::vector<float>(), UV);
                for (int32_t v : quad) m.vi.push_back((uint32_t)v);
                a_->add_patch_mesh(std::move(m), (uint32_t)gs_.material, gs_.area_light, object_);
            }
        } else {
            fail(tk.where(line) + ": shape \"" + type + "\" is not supported by this backend (sphere, trianglemesh, bilinearmesh, plymesh)", SHM_ERR_UNSUPPORTED);
        }
    }
    std::string resolve(const std::string& f) const { return (f.empty() || f[0] == '/' || base_dir_.empty()) ? f : base_dir_ + "/" + f; }

    // ---- lights (light.rs:92-200, 424-452) ----
    void light_source(const std::string& type, const Params& ps, Tokenizer& tk, int line) {
        ShmLight l;
        memset(&l, 0, sizeof(l));
        float scale = ps.one_float("scale", 1.0f);
        if (type == "point") {
            l.kind = SHM_LIGHT_POINT;
            SpectrumValue d65;  // light.rs:433: the dictionary's colour space illuminant when "I" is absent
            d65.kind = SpectrumValue::DENSE;
            d65.dense = illuminant_dense(gs_.color_space);
            const Param* ip = ps.find("I");
            const std::vector<float> dense = a_->dense_of(ip ? spectrum_of(*ip, tk) : d65);
            scale /= spectrum_to_photometric(dense);
            const float phi_v = ps.one_float("power", -1.0f);
            if (phi_v > 0.0f) scale *= phi_v / (4.0f * 3.14159265358979323846f);
            const std::vector<float> from = ps.floats("from");
            const Xf rfl = xf_mul(render_from_object(), xf_translate(from.size() == 3 ? from[0] : 0.0f, from.size() == 3 ? from[1] : 0.0f, from.size() == 3 ? from[2] : 0.0f));
            const V3 pos = xf_point(rfl.m, shm::v3s(0.0f));
            l.position[0] = pos.x; l.position[1] = pos.y; l.position[2] = pos.z;
            l.scale = scale;
            l.spectrum = a_->spec_dense(dense);
        } else if (type == "infinite" && !ps.one_string("filename", "").empty()) {  // light.rs:147-236: ImageInfinitelight
            if (ps.find("portal")) fail(tk.where(line) + ": portal infinite lights are todo!() in the reference", SHM_ERR_UNSUPPORTED);
            if (ps.find("L")) fail(tk.where(line) + ": Can't specify both emission L and filename with ImageInfinitelight");
            const std::string file = resolve(ps.one_string("filename", ""));
            HostImage image;
            try { image = image_read(file, ColorEncoding::get("srgb")); }  // Image::read(path, None): the sRGB encoding (image.rs:1152-1156)
            catch (const LoadError& e) { fail(tk.where(line) + ": " + e.what(), e.code); }
            if (!image.srgb_color_space) fail(tk.where(line) + ": " + file + ": Expected color space (a grey PNG carries none, light.rs:169)");
            if (image.nc < 3) fail(tk.where(line) + ": Infinite image light sources must have RGB channels");
            if (image.res[0] != image.res[1]) fail(tk.where(line) + ": " + file + ": image resolution is non-square; it's unlikely that it is an equal-area environment map (light.rs:918-923)");
            a_->need_color_space();
            scale /= spectrum_to_photometric(a_->cs_illuminant());  // the IMAGE's colour space (light.rs:178-188): sRGB for every RGB PNG
            const float e_v = ps.one_float("illuminance", -1.0f);
            if (e_v > 0.0f) {  // light.rs:191-221: the upper hemisphere's illuminance of the map
                float lum[3];
                a_->srgb_luminance_vector(lum);
                float illuminance = 0.0f;
                for (int y = 0; y < image.res[1]; ++y) {
                    const float v = ((float)y + 0.5f) / (float)image.res[1];
                    for (int x = 0; x < image.res[0]; ++x) {
                        const float u = ((float)x + 0.5f) / (float)image.res[0];
                        const V3 w = shm::equal_area_square_to_sphere(shm::v2(u, v));
                        if (w.z <= 0.0f) continue;
                        for (int c = 0; c < 3; ++c) illuminance += image.get(x, y, c) * lum[c] * w.z;
                    }
                }
                illuminance *= 2.0f * 3.14159265358979323846f / (float)(image.res[0] * image.res[1]);
                scale *= e_v / illuminance;
            }
            ShmImageInfiniteLight il;
            memset(&il, 0, sizeof(il));
            const Xf rfl = render_from_object();
            memcpy(il.render_from_light, rfl.m.m, sizeof(float) * 16);
            memcpy(il.light_from_render, rfl.inv.m, sizeof(float) * 16);
            il.image_level = (uint32_t)a_->image_levels.size();
            ShmImageLevel lv;
            memset(&lv, 0, sizeof(lv));
            lv.width = image.res[0]; lv.height = image.res[1];
            lv.texel_offset = (uint32_t)a_->texels.size();
            a_->image_levels.push_back(lv);
            for (int y = 0; y < image.res[1]; ++y) for (int x = 0; x < image.res[0]; ++x) for (int c = 0; c < 3; ++c) a_->texels.push_back(image.get(x, y, c));  // select_channels(R, G, B)
            a_->image_lights.push_back(il);
            l.kind = SHM_LIGHT_IMAGE_INFINITE;
            l.primitive = (uint32_t)a_->image_lights.size() - 1;
            l.scale = scale;
        } else if (type == "infinite") {
            if (ps.find("portal")) fail(tk.where(line) + ": portal infinite lights are todo!() in the reference", SHM_ERR_UNSUPPORTED);
            l.kind = SHM_LIGHT_UNIFORM_INFINITE;
            SpectrumValue d65;  // light.rs:114, 141-145: color_space.illuminant
            d65.kind = SpectrumValue::DENSE;
            d65.dense = illuminant_dense(gs_.color_space);
            const Param* lp = ps.find("L");
            const std::vector<float> dense = a_->dense_of(lp ? spectrum_of(*lp, tk) : d65);
            scale /= spectrum_to_photometric(dense);
            const float e_v = ps.one_float("illuminance", -1.0f);
            if (e_v > 0.0f) scale *= e_v / (4.0f * 3.14159265358979323846f);  // light.rs:128-133 (k_e = 4 pi there)
            l.scale = scale;
            l.spectrum = a_->spec_dense(dense);
        } else {
            fail(tk.where(line) + ": light \"" + type + "\" is not supported by this backend (point, infinite)", SHM_ERR_UNSUPPORTED);
        }
        a_->lights.push_back(l);
    }

    // ---- WorldBegin: camera, film, filter become concrete (scene.rs:84-145, camera.rs, film.rs) ----
    void world_begin(Tokenizer& tk, int line) {
        ShmFilm& f = a_->film;
        if (film_type_ != "rgb") fail(tk.where(line) + ": film \"" + film_type_ + "\" is not supported (rgb)", SHM_ERR_UNSUPPORTED);
        const int xres = film_params_.one_int("xresolution", 1280), yres = film_params_.one_int("yresolution", 720);
        if (xres <= 0 || yres <= 0) fail("film resolution must be positive");
        f.full_resolution[0] = xres; f.full_resolution[1] = yres;
        int pb[4] = {0, 0, xres, yres};
        const std::vector<int> pbv = film_params_.ints("pixelbounds");
        if (!pbv.empty()) {
            if (pbv.size() != 4) fail("Too many values supplied for pixel bounds, expected 4");
            pb[0] = std::max(0, pbv[0]); pb[2] = std::min(xres, pbv[1]); pb[1] = std::max(0, pbv[2]); pb[3] = std::min(yres, pbv[3]);  // (x0, x1, y0, y1), film.rs:281
            if (pb[2] <= pb[0] || pb[3] <= pb[1]) fail("Supplied bounds do not intersect with image!");
        }
        const std::vector<float> cr = film_params_.floats("cropwindow");
        if (!cr.empty()) {
            if (cr.size() != 4) fail("cropwindow expects four values");
            pb[0] = (int)std::ceil(xres * std::min(std::max(cr[0], 0.0f), 1.0f)); pb[2] = (int)std::ceil(xres * std::min(std::max(cr[1], 0.0f), 1.0f));
            pb[1] = (int)std::ceil(yres * std::min(std::max(cr[2], 0.0f), 1.0f)); pb[3] = (int)std::ceil(yres * std::min(std::max(cr[3], 0.0f), 1.0f));
            if (pb[2] <= pb[0] || pb[3] <= pb[1]) fail("Degenerate pixel bounds provided to film");
        }
        memcpy(f.pixel_bounds, pb, sizeof(pb));
        // PixelSensor::create (film.rs:767-818): a named camera sensor needs "<name>_r/_g/_b" spectra that NamedSpectrum does not hold — the
        // reference panics on every name but cie1931; "whitebalance" is a colour temperature for the von Kries matrix of the output transform
        if (film_params_.one_string("sensor", "cie1931") != "cie1931")
            fail(tk.where(line) + ": Unknown sensor type \"" + film_params_.one_string("sensor", "") + "\" (the reference knows cie1931 only)");
        settings_.white_balance = film_params_.one_float("whitebalance", 0.0f);
        const float shutter_open = camera_params_.one_float("shutteropen", 0.0f), shutter_close = camera_params_.one_float("shutterclose", 1.0f);
        f.imaging_ratio = (shutter_close - shutter_open) * film_params_.one_float("iso", 100.0f) / 100.0f;  // film.rs:785
        f.max_component_value = film_params_.one_float("maxcomponentvalue", INFINITY);
        settings_.filename = film_params_.one_string("filename", "shimmer.pfm");
        if (filter_type_ != "box") fail(tk.where(line) + ": pixel filter \"" + filter_type_ + "\" is not supported (box)", SHM_ERR_UNSUPPORTED);
        f.filter_radius[0] = filter_params_.one_float("xradius", 0.5f);  // filter.rs:70-80
        f.filter_radius[1] = filter_params_.one_float("yradius", 0.5f);
        // camera: world_from_camera = inverse(CTM at the Camera directive); render space = CameraWorld (camera.rs:507-523)
        const Xf world_from_camera = xf_inverse(camera_from_world_);
        float wfc[16], rfw[16];
        memcpy(wfc, world_from_camera.m.m, sizeof(wfc));
        const float lens = camera_params_.one_float("lensradius", 0.0f), focal = camera_params_.one_float("focaldistance", 1e6f);
        if (camera_type_ != "perspective" && camera_type_ != "orthographic")
            fail(tk.where(line) + ": camera \"" + camera_type_ + "\" is not supported (perspective, orthographic)", SHM_ERR_UNSUPPORTED);
        ShmCameraParams cp;
        memset(&cp, 0, sizeof(cp));
        cp.kind = camera_type_ == "perspective" ? SHM_CAMERA_PERSPECTIVE : SHM_CAMERA_ORTHOGRAPHIC;
        cp.render_space = settings_.render_space;
        memcpy(cp.world_from_camera, wfc, sizeof(wfc));
        cp.fov_deg = camera_params_.one_float("fov", 90.0f);
        cp.full_resolution[0] = xres; cp.full_resolution[1] = yres;
        cp.lens_radius = lens; cp.focal_distance = focal;
        cp.frame_aspect_ratio = camera_params_.one_float("frameaspectratio", 0.0f);
        const std::vector<float> sw = camera_params_.floats("screenwindow");
        if (sw.size() == 4) { cp.has_screen_window = 1; memcpy(cp.screen_window, sw.data(), 16); }  // (any other count: the reference warns and ignores it)
        if (shm_camera_create(&cp, &a_->camera, rfw) != SHM_OK) fail(tk.where(line) + ": camera construction failed (fov, screen window or transform)");
        a_->camera.shutter_open = shutter_open;
        a_->camera.shutter_close = shutter_close;
        a_->have_camera = true;
        memcpy(render_from_world_.m.m, rfw, sizeof(rfw));
        if (!m4_inverse(render_from_world_.m, render_from_world_.inv)) fail("singular render_from_world");
        // scene.rs:1659-1672
        gs_.ctm = xf_identity();
        coordinate_systems_["world"] = gs_.ctm;
        world_ = true;
    }

    void directive(const Token& t, Tokenizer& tk) {
        const std::string& d = t.text;
        float v[16];
        if (d == "Identity") gs_.ctm = xf_identity();
        else if (d == "Translate") { read_floats(tk, v, 3); gs_.ctm = xf_mul(gs_.ctm, xf_translate(v[0], v[1], v[2])); }
        else if (d == "Scale") { read_floats(tk, v, 3); gs_.ctm = xf_mul(gs_.ctm, xf_scale(v[0], v[1], v[2])); }
        else if (d == "Rotate") { read_floats(tk, v, 4); gs_.ctm = xf_mul(gs_.ctm, xf_rotate(v[0], shm::v3(v[1], v[2], v[3]))); }
        else if (d == "LookAt") {
            read_floats(tk, v, 9);
            Xf la;
            if (!xf_look_at(shm::v3(v[0], v[1], v[2]), shm::v3(v[3], v[4], v[5]), shm::v3(v[6], v[7], v[8]), la)) fail(tk.where(t.line) + ": Uninvertible look_at!");
            gs_.ctm = xf_mul(gs_.ctm, la);
        } else if (d == "Transform" || d == "ConcatTransform") {  // scene.rs:1509-1526: the file holds the matrix column-major
            read_floats(tk, v, 16);
            Xf x;
            M4 given;
            memcpy(given.m, v, sizeof(v));
            x.m = m4_transpose(given);
            if (!m4_inverse(x.m, x.inv)) fail(tk.where(t.line) + ": singular matrix");
            gs_.ctm = d == "Transform" ? x : xf_mul(gs_.ctm, x);
        } else if (d == "CoordinateSystem") coordinate_systems_[read_string(tk, t)] = gs_.ctm;
        else if (d == "CoordSysTransform") {
            const std::string n = read_string(tk, t);
            auto it = coordinate_systems_.find(n);
            if (it != coordinate_systems_.end()) gs_.ctm = it->second;  // (the reference only warns when the name is unknown)
        } else if (d == "ReverseOrientation") gs_.reverse_orientation = !gs_.reverse_orientation;
        else if (d == "TransformTimes" || d == "ActiveTransform") fail(tk.where(t.line) + ": animated transforms are not supported", SHM_ERR_UNSUPPORTED);
        else if (d == "ColorSpace") {  // scene.rs:1561-1564: RgbColorSpace::get_named(n.into()) — srgb / rec2020 / aces2065-1, any case (colorspace.rs:123-132)
            const std::string n = read_string(tk, t);
            const int cs = color_space_from_name(n);
            if (cs < 0) fail(tk.where(t.line) + ": Unknown color space: " + n);
            gs_.color_space = cs;
        }
        else if (d == "Option") {  // options.rs / scene.rs:1375-1454
            const Params ps = read_params(tk);
            for (const Param& p : ps.v) {
                if (p.name == "seed") settings_.seed = p.i.empty() ? 0 : p.i[0];
                else if (p.name == "disablepixeljitter") settings_.disable_pixel_jitter = !p.b.empty() && p.b[0];
                else if (p.name == "disablewavelengthjitter") settings_.disable_wavelength_jitter = !p.b.empty() && p.b[0];
                else if (p.name == "forcediffuse") settings_.force_diffuse = !p.b.empty() && p.b[0];
                else if (p.name == "disabletexturefiltering") settings_.disable_texture_filtering = !p.b.empty() && p.b[0];
                else if (p.name == "rendercoordsys") {  // scene.rs:1411-1431
                    const std::string v = p.s.empty() ? "" : p.s[0];
                    if (v == "cameraworld") settings_.render_space = SHM_RENDER_SPACE_CAMERA_WORLD;
                    else if (v == "camera") settings_.render_space = SHM_RENDER_SPACE_CAMERA;
                    else if (v == "world") settings_.render_space = SHM_RENDER_SPACE_WORLD;
                    else fail(tk.where(t.line) + ": Unknown rendering coordinate system " + v);
                }
            }
        } else if (d == "Camera") {
            need_world(t, tk, false);
            camera_type_ = read_string(tk, t);
            camera_params_ = read_params(tk);
            camera_from_world_ = gs_.ctm;                       // scene.rs:1614-1637
            coordinate_systems_["camera"] = xf_inverse(gs_.ctm);
        } else if (d == "Film") { need_world(t, tk, false); film_type_ = read_string(tk, t); film_params_ = read_params(tk); film_color_space_ = gs_.color_space; }  // scene.rs:95: film.parameters.color_space
        else if (d == "PixelFilter") { need_world(t, tk, false); filter_type_ = read_string(tk, t); filter_params_ = read_params(tk); }
        else if (d == "Sampler") {
            need_world(t, tk, false);
            settings_.sampler = read_string(tk, t);
            const Params ps = read_params(tk);
            if (settings_.sampler != "independent") fail(tk.where(t.line) + ": sampler \"" + settings_.sampler + "\" is not supported (independent)", SHM_ERR_UNSUPPORTED);
            settings_.spp = ps.one_int("pixelsamples", 4);
            settings_.seed = ps.one_int("seed", settings_.seed);
        } else if (d == "Integrator") {
            need_world(t, tk, false);
            settings_.integrator = read_string(tk, t);
            const Params ps = read_params(tk);
            if (settings_.integrator != "path" && settings_.integrator != "simplepath" && settings_.integrator != "randomwalk") fail(tk.where(t.line) + ": Unknown integrator " + settings_.integrator);
            settings_.max_depth = ps.one_int("maxdepth", 5);
            settings_.regularize = ps.one_bool("regularize", false);
            settings_.sample_lights = ps.one_bool("samplelights", true);
            settings_.sample_bsdf = ps.one_bool("samplebsdf", true);
            if (ps.one_string("lightsampler", "uniform") != "uniform") fail(tk.where(t.line) + ": only the uniform light sampler exists on this path", SHM_ERR_UNSUPPORTED);
        } else if (d == "Accelerator") { read_string(tk, t); read_params(tk); }  // the BVH of aggregate.rs is the only accelerator
        else if (d == "WorldBegin") { need_world(t, tk, false); world_begin(tk, t.line); }
        else if (d == "AttributeBegin") { need_world(t, tk, true); stack_.push_back(gs_); push_stack_.push_back(Pushed{'a', tk.where(t.line)}); }
        else if (d == "AttributeEnd") {  // scene.rs:1693-1712
            if (push_stack_.empty() || stack_.empty()) fail(tk.where(t.line) + ": Unmatched attribute_end statement.");
            if (push_stack_.back().kind == 'o') fail(tk.where(t.line) + ": Mismatched nesting: open ObjectBegin from " + push_stack_.back().where + " at attribute_end.");
            gs_ = stack_.back();
            stack_.pop_back();
            push_stack_.pop_back();
        } else if (d == "Attribute") {  // scene.rs:1714-1730: default parameters for what follows of that kind, within the attribute scope
            const std::string target = read_string(tk, t);
            const Params ps = read_params(tk);
            Params* dst = target == "shape" ? &gs_.shape_attributes : target == "light" ? &gs_.light_attributes : target == "material" ? &gs_.material_attributes
                          : target == "texture" ? &gs_.texture_attributes : nullptr;
            if (target == "medium") fail(tk.where(t.line) + ": participating media are todo!() in the reference and not supported", SHM_ERR_UNSUPPORTED);
            if (!dst) fail(tk.where(t.line) + ": Unknown attribute target " + target);
            dst->v.insert(dst->v.end(), ps.v.begin(), ps.v.end());
        }
        else if (d == "Material") { need_world(t, tk, true); const std::string ty = read_string(tk, t); const Params ps = with_attributes(read_params(tk), gs_.material_attributes); gs_.material = make_material(ty, ps, &tk); }
        else if (d == "MakeNamedMaterial") {
            need_world(t, tk, true);
            const std::string name = read_string(tk, t);
            const Params ps = with_attributes(read_params(tk), gs_.material_attributes);
            if (named_materials_.count(name)) fail(tk.where(t.line) + ": named material \"" + name + "\" redefined");
            const std::string ty = ps.one_string("type", "");
            if (ty.empty()) fail(tk.where(t.line) + ": MakeNamedMaterial \"" + name + "\" has no \"string type\"");
            named_materials_[name] = make_material(ty, ps, &tk);
        } else if (d == "NamedMaterial") {
            need_world(t, tk, true);
            const std::string name = read_string(tk, t);
            auto it = named_materials_.find(name);
            if (it == named_materials_.end()) fail(tk.where(t.line) + ": " + name + ": named material not found.");
            gs_.material = it->second;
        } else if (d == "Texture") {
            need_world(t, tk, true);
            const std::string name = read_string(tk, t), ty = read_string(tk, t), cls = read_string(tk, t);
            texture(name, ty, cls, with_attributes(read_params(tk), gs_.texture_attributes), tk, t.line);
        } else if (d == "AreaLightSource") {
            need_world(t, tk, true);
            const std::string ty = read_string(tk, t);
            const Params ps = with_attributes(read_params(tk), gs_.light_attributes);
            if (ty != "diffuse") fail(tk.where(t.line) + ": area light \"" + ty + "\" unknown (diffuse)");
            if (!ps.one_string("filename", "").empty()) fail(tk.where(t.line) + ": image area lights are todo!() in the reference", SHM_ERR_UNSUPPORTED);
            Assembly::Emission em;
            em.on = true;
            const Param* lp = ps.find("L");
            if (lp) em.L = spectrum_of(*lp, tk);
            else { em.L.kind = SpectrumValue::DENSE; em.L.dense = illuminant_dense(gs_.color_space); em.L.key = color_space_def(gs_.color_space).aces_d60 ? "illum-acesD60" : "StdIllum-D65"; }  // light.rs:592-596: color_space.illuminant
            em.scale = ps.one_float("scale", 1.0f);
            em.power = ps.one_float("power", -1.0f);
            em.two_sided = ps.one_bool("twosided", false);
            gs_.area_light = em;
        } else if (d == "LightSource") { need_world(t, tk, true); const std::string ty = read_string(tk, t); light_source(ty, with_attributes(read_params(tk), gs_.light_attributes), tk, t.line); }
        else if (d == "Shape") { need_world(t, tk, true); const std::string ty = read_string(tk, t); shape(ty, with_attributes(read_params(tk), gs_.shape_attributes), tk, t.line); }
        else if (d == "ObjectBegin") {  // scene.rs:1904-1982
            need_world(t, tk, true);
            const std::string name = read_string(tk, t);
            if (object_ != 0) fail(tk.where(t.line) + ": ObjectBegin called inside of instance definition");
            if (a_->objects.count(name)) fail(tk.where(t.line) + ": " + name + ": trying to redefine an object instance");
            stack_.push_back(gs_);
            push_stack_.push_back(Pushed{'o', tk.where(t.line)});
            a_->objects[name] = (uint32_t)a_->objects.size() + 1;
            object_ = a_->objects[name];
        } else if (d == "ObjectEnd") {
            if (object_ == 0) fail(tk.where(t.line) + ": ObjectEnd called outside of instance definition");
            if (push_stack_.empty() || stack_.empty()) fail(tk.where(t.line) + ": Unmatched ObjectEnd statement.");  // scene.rs:1947-1950
            if (push_stack_.back().kind == 'a') fail(tk.where(t.line) + ": Mismatched nesting: open AttributeBegin from " + push_stack_.back().where + " at ObjectEnd.");
            gs_ = stack_.back();
            stack_.pop_back();
            push_stack_.pop_back();
            object_ = 0;
        } else if (d == "ObjectInstance") {
            need_world(t, tk, true);
            const std::string name = read_string(tk, t);
            if (object_ != 0) fail(tk.where(t.line) + ": ObjectInstance can't be called inside instance definition");
            auto it = a_->objects.find(name);
            if (it == a_->objects.end()) fail(tk.where(t.line) + ": " + name + ": object instance not defined");
            // The object's shapes were moved to render space with THEIR ctm; the instance applies render_from_world * ctm * world_from_render
            // (scene.rs:1984-2006: render_from_instance = render_from_object * world_from_render)
            const Xf rfi = xf_mul(render_from_object(), xf_inverse(render_from_world_));
            a_->instances.push_back(Assembly::Inst{it->second, rfi.m});
            Assembly::Prim pr{SHM_SHAPE_INSTANCE, (uint32_t)a_->instances.size() - 1, 0u, -1, 0u, {0}};
            a_->prims.push_back(pr);
        } else if (d == "Include") {
            const std::string fn = resolve(read_string(tk, t));
            // the reference recurses without a limit (parser.rs:191-197); a file that includes itself would overflow the native stack here,
            // which the try / catch at the ABI boundary cannot turn into an error code
            if (std::find(include_chain_.begin(), include_chain_.end(), fn) != include_chain_.end()) fail(tk.where(t.line) + ": Include cycle: " + fn + " is already being parsed");
            if (include_chain_.size() >= 64) fail(tk.where(t.line) + ": Include nesting deeper than 64 files");
            std::ifstream in(fn);
            if (!in) fail(tk.where(t.line) + ": unable to read included file " + fn);
            std::stringstream ss;
            ss << in.rdbuf();
            parse(ss.str(), fn);
        } else if (d == "Import") { const std::string fn = read_string(tk, t); fail(tk.where(t.line) + ": Import \"" + fn + "\" is not supported (todo!() in the reference: use Include)", SHM_ERR_UNSUPPORTED); }
        else if (d == "MakeNamedMedium" || d == "MediumInterface") fail(tk.where(t.line) + ": participating media are todo!() in the reference and not supported", SHM_ERR_UNSUPPORTED);
        else fail(tk.where(t.line) + ": unknown directive \"" + d + "\"");
    }
};

struct Holder {
    std::unique_ptr<Assembly::Built> built;
};

static int load_text(const std::string& text, const std::string& name, const std::string& base_dir, ShmPbrtScene** out) {
    if (!out) return SHM_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    try {
        Loader loader(base_dir);
        loader.parse(text, name);
        const Loader::Settings st = loader.settings_;
        std::unique_ptr<Assembly::Built> built = loader.finish();
        std::unique_ptr<ShmPbrtScene> scene(new ShmPbrtScene());
        memset(scene.get(), 0, sizeof(ShmPbrtScene));
        scene->desc = built->desc;
        ShmRenderParams& p = scene->params;
        p.seed = (uint64_t)(int64_t)st.seed;
        p.samples_per_pixel = st.spp;
        p.max_depth = st.max_depth;
        p.regularize = st.regularize;
        p.disable_pixel_jitter = st.disable_pixel_jitter;
        p.disable_wavelength_jitter = st.disable_wavelength_jitter;
        p.force_diffuse = st.force_diffuse;
        p.disable_texture_filtering = st.disable_texture_filtering;
        p.sample_lights = st.sample_lights;
        p.sample_bsdf = st.sample_bsdf;
        // SHM_REFERENCE_QUIRKS (include/shimmer_hip.h ShmRenderParams::disable_reference_quirks): ON unless the host's environment says 0 / off
        if (const char* q = getenv("SHM_REFERENCE_QUIRKS")) p.disable_reference_quirks = (!strcmp(q, "0") || !strcmp(q, "off") || !strcmp(q, "OFF")) ? 1 : 0;
        p.integrator = st.integrator == "path" ? SHM_INTEGRATOR_PATH : (st.integrator == "simplepath" ? SHM_INTEGRATOR_SIMPLE_PATH : SHM_INTEGRATOR_RANDOM_WALK);
        snprintf(scene->integrator, sizeof(scene->integrator), "%s", st.integrator.c_str());
        snprintf(scene->output_filename, sizeof(scene->output_filename), "%s", st.filename.c_str());
        Assembly::film_output_matrix(st.white_balance, scene->output_rgb_from_sensor_rgb, loader.film_color_space());
        Holder* h = new Holder();
        h->built = std::move(built);
        scene->owner = h;
        *out = scene.release();
        return SHM_OK;
    } catch (const LoadError& e) {
        shm_set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        shm_set_last_error("out of memory while loading the scene");
        return SHM_ERR_OUT_OF_MEMORY;
    } catch (const std::exception& e) {  // nothing unwinds across the ABI
        shm_set_last_error(e.what());
        return SHM_ERR_INTERNAL;
    }
}

}  // namespace pbrt

extern "C" {

int shm_scene_parse_pbrt(const char* text, const char* base_dir, ShmPbrtScene** out) {
    if (!text) return SHM_ERR_INVALID_ARGUMENT;
    return pbrt::load_text(text, "<string>", base_dir ? base_dir : "", out);
}

int shm_scene_load_pbrt(const char* path, ShmPbrtScene** out) {
    if (!path || !out) return SHM_ERR_INVALID_ARGUMENT;
    std::ifstream in(path);
    if (!in) { shm_set_last_error((std::string("unable to read ") + path).c_str()); return SHM_ERR_INVALID_ARGUMENT; }
    std::stringstream ss;
    ss << in.rdbuf();
    std::string p(path);
    const size_t slash = p.find_last_of('/');
    return pbrt::load_text(ss.str(), p, slash == std::string::npos ? std::string(".") : p.substr(0, slash), out);
}

// ---- test entries: the product's own tokenizer and parameter-list parser, so that the reference's in-source vectors for them
// (loading/tokenizer.rs:126-260, token.rs:218-290, param.rs:214-260, parser.rs:656-870) can be replayed against this code ----
int shm_pbrt_tokenize(const char* text, char* out, uint64_t capacity, uint32_t* n_tokens) {
    if (!text || !out || !n_tokens) return SHM_ERR_INVALID_ARGUMENT;
    try {
        const std::string s(text);
        pbrt::Tokenizer tk(s, "<string>");
        uint64_t used = 0;
        uint32_t n = 0;
        for (;;) {
            const pbrt::Token t = tk.next_raw();
            if (t.kind == pbrt::Token::END) break;
            // kind letter, then the token as the reference's Token holds it, then NUL
            const char kind = t.kind == pbrt::Token::WORD ? (pbrt::is_directive_name(t.text) ? 'D' : 'W') : t.kind == pbrt::Token::STRING ? 'S'
                              : t.kind == pbrt::Token::BAD_QUOTE ? 'Q' : 'B';
            const std::string raw = t.raw();
            if (used + raw.size() + 2 > capacity) { shm_set_last_error("shm_pbrt_tokenize: output buffer too small"); return SHM_ERR_INVALID_ARGUMENT; }
            out[used++] = kind;
            memcpy(out + used, raw.data(), raw.size());
            used += raw.size();
            out[used++] = 0;
            ++n;
        }
        *n_tokens = n;
        return SHM_OK;
    } catch (const std::exception& e) {
        shm_set_last_error(e.what());
        return SHM_ERR_INVALID_ARGUMENT;
    }
}

int shm_pbrt_parse_params(const char* text, char* out_json, uint64_t capacity) {
    if (!text || !out_json || capacity == 0) return SHM_ERR_INVALID_ARGUMENT;
    try {
        const std::string s(text);
        pbrt::Tokenizer tk(s, "<string>");
        const pbrt::Params ps = pbrt::parse_params(tk);
        if (tk.peek().kind != pbrt::Token::END) pbrt::fail(tk.where(tk.peek().line) + ": trailing tokens after the parameter list");
        std::ostringstream o;
        o << "[";
        for (size_t k = 0; k < ps.v.size(); ++k) {
            const pbrt::Param& p = ps.v[k];
            o << (k ? "," : "") << "{\"type\":\"" << p.type << "\",\"name\":\"" << p.name << "\",\"floats\":[";
            for (size_t i = 0; i < p.f.size(); ++i) { char b[40]; snprintf(b, sizeof(b), "%.9g", (double)p.f[i]); o << (i ? "," : "") << b; }
            o << "],\"ints\":[";
            for (size_t i = 0; i < p.i.size(); ++i) o << (i ? "," : "") << p.i[i];
            o << "],\"bools\":[";
            for (size_t i = 0; i < p.b.size(); ++i) o << (i ? "," : "") << (p.b[i] ? "true" : "false");
            o << "],\"strings\":[";
            for (size_t i = 0; i < p.s.size(); ++i) {
                o << (i ? "," : "") << "\"";
                for (char c : p.s[i]) { if (c == '"' || c == '\\') o << '\\'; if (c == '\n') o << "\\n"; else o << c; }
                o << "\"";
            }
            o << "]}";
        }
        o << "]";
        const std::string j = o.str();
        if (j.size() + 1 > capacity) { shm_set_last_error("shm_pbrt_parse_params: output buffer too small"); return SHM_ERR_INVALID_ARGUMENT; }
        memcpy(out_json, j.c_str(), j.size() + 1);
        return SHM_OK;
    } catch (const pbrt::LoadError& e) {
        shm_set_last_error(e.what());
        return e.code;
    } catch (const std::exception& e) {
        shm_set_last_error(e.what());
        return SHM_ERR_INTERNAL;
    }
}

void shm_pbrt_free(ShmPbrtScene* scene) {
    if (!scene) return;
    delete static_cast<pbrt::Holder*>(scene->owner);
    delete scene;
}

int shm_blackbody_dense(float temperature_kelvin, float out471[471]) {
    if (!out471 || !(temperature_kelvin > 0.0f)) return SHM_ERR_INVALID_ARGUMENT;
    const std::vector<float> d = pbrt::blackbody_dense(temperature_kelvin);
    memcpy(out471, d.data(), sizeof(float) * 471);
    return SHM_OK;
}

int shm_image_load_png(const char* path, const char* encoding, uint32_t wrap, int build_pyramid, ShmLoadedImage* out) {
    if (!path || !out || wrap > SHM_WRAP_OCTAHEDRAL_SPHERE) return SHM_ERR_INVALID_ARGUMENT;
    memset(out, 0, sizeof(*out));
    try {
        const pbrt::HostImage image = pbrt::image_read(path, pbrt::ColorEncoding::get(encoding ? encoding : "sRGB"));
        std::vector<pbrt::HostImage> levels;
        if (build_pyramid) levels = pbrt::generate_pyramid(pbrt::mipmap_select_channels(image), (pbrt::WrapMode)wrap);
        else levels.push_back(image);
        const int keep = levels[0].nc >= 3 ? 3 : 1;
        out->n_levels = (uint32_t)levels.size();
        out->n_channels = (uint32_t)keep;
        out->file_channels = (uint32_t)levels[0].nc;
        out->has_color_space = image.srgb_color_space ? 1u : 0u;
        uint64_t total = 0;
        for (const pbrt::HostImage& lv : levels) total += (uint64_t)lv.res[0] * (uint64_t)lv.res[1] * (uint64_t)keep;
        if (total > 0xffffffffull) pbrt::fail("image pyramid: more than 2^32 texel floats");
        out->levels = static_cast<ShmImageLevel*>(calloc(levels.size(), sizeof(ShmImageLevel)));
        out->texels = static_cast<float*>(malloc(sizeof(float) * (size_t)total));
        if (!out->levels || !out->texels) { shm_image_free(out); return SHM_ERR_OUT_OF_MEMORY; }
        uint64_t k = 0;
        for (size_t i = 0; i < levels.size(); ++i) {
            const pbrt::HostImage& lv = levels[i];
            out->levels[i].width = lv.res[0];
            out->levels[i].height = lv.res[1];
            out->levels[i].texel_offset = (uint32_t)k;
            for (int y = 0; y < lv.res[1]; ++y) for (int x = 0; x < lv.res[0]; ++x) for (int c = 0; c < keep; ++c) out->texels[k++] = lv.get(x, y, c);
        }
        out->n_texel_floats = total;
        return SHM_OK;
    } catch (const pbrt::LoadError& e) {
        shm_image_free(out);
        shm_set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        shm_image_free(out);
        shm_set_last_error("out of memory while reading the image");
        return SHM_ERR_OUT_OF_MEMORY;
    } catch (const std::exception& e) {
        shm_image_free(out);
        shm_set_last_error(e.what());
        return SHM_ERR_INTERNAL;
    }
}

void shm_image_free(ShmLoadedImage* image) {
    if (!image) return;
    free(image->levels);
    free(image->texels);
    memset(image, 0, sizeof(*image));
}

int shm_look_at(const float eye[3], const float look[3], const float up[3], float world_from_camera_out[16]) {
    if (!eye || !look || !up || !world_from_camera_out) return SHM_ERR_INVALID_ARGUMENT;
    pbrt::M4 wfc;
    if (!pbrt::look_at_world_from_camera(shm::v3(eye[0], eye[1], eye[2]), shm::v3(look[0], look[1], look[2]), shm::v3(up[0], up[1], up[2]), wfc)) {
        shm_set_last_error("Uninvertible look_at!");
        return SHM_ERR_INVALID_ARGUMENT;
    }
    memcpy(world_from_camera_out, wfc.m, sizeof(float) * 16);
    return SHM_OK;
}

}  // extern "C"
