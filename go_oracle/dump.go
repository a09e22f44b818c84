// dump.go — closes the loop this repository cannot close by itself: the image the reference's Go implementation writes for a token list
// (NewVocab + Save, go/tokenmonster.go:2742, :2602) and the ids it produces (Tokenize / Count, :959, :971), in the JSON shape of
// tests/golden/*.json.  The development image of libtokenmonster_hip.so has no Go toolchain and no network (the reference's
// dependencies are neither vendored nor pinned), so nobody has run this there; tests/test_go_fixtures.py consumes the files it writes
// whenever they are present and compares
//   * tm_build_vocab's image of the same token list, byte for byte, with `vocab_b64` (the rules of go/tokenmonster.go:3423-3793), and
//   * the oracle's and the HIP path's ids of the same documents with `ids` / `missing` / `count`.
//
// With Go and network access (the steps of training/README.md:38-45):
//     python go_oracle/make_cases.py > go_oracle/cases.json          # here: the seeded token lists of tests/golden/make_builder_golden.py
//     cd go_oracle && go mod init go_oracle && go mod tidy && go run dump.go cases.json ../tests/golden
//     python -m pytest tests/test_go_fixtures.py
package main

import (
	"encoding/base64"
	"encoding/json"
	"fmt"
	"os"
	"path/filepath"

	"github.com/alasdairforsythe/tokenmonster"
)

type caseIn struct {
	Name      string   `json:"name"`
	TokensB64 []string `json:"tokens_b64"`
	Special   []int    `json:"special"`        // 1 = the token of the same position is a special token (may be empty)
	Capcode   uint8    `json:"capcode"`
	Charset   uint8    `json:"charset"`        // 1 UTF-8, 2 UTF-16
	Normalize string   `json:"normalization"`  // "" or e.g. "NFD"
	DocsB64   []string `json:"docs_b64"`       // ALREADY NORMALIZED documents are not possible through the public API: these are raw documents
}

type caseOut struct {
	Note      string     `json:"note"`
	Generator string     `json:"generator"`
	Name      string     `json:"name"`
	TokensB64 []string   `json:"tokens_b64"`
	Special   []int      `json:"special"`
	Capcode   uint8      `json:"capcode"`
	Charset   uint8      `json:"charset"`
	Normalize string     `json:"normalization"`
	VocabB64  string     `json:"vocab_b64"`
	DocsB64   []string   `json:"docs_b64"`
	Ids       [][]uint32 `json:"ids"`
	Missing   []int      `json:"missing"`
	Count     []int      `json:"count"`
}

func must(err error) {
	if err != nil {
		fmt.Fprintln(os.Stderr, "dump:", err)
		os.Exit(1)
	}
}

func main() {
	if len(os.Args) != 3 {
		fmt.Fprintln(os.Stderr, "usage: go run dump.go cases.json <output directory>")
		os.Exit(2)
	}
	raw, err := os.ReadFile(os.Args[1])
	must(err)
	var cases []caseIn
	must(json.Unmarshal(raw, &cases))
	for _, c := range cases {
		var tokens, special [][]byte
		for i, t := range c.TokensB64 {
			b, err := base64.StdEncoding.DecodeString(t)
			must(err)
			if i < len(c.Special) && c.Special[i] != 0 {
				special = append(special, b)
			} else {
				tokens = append(tokens, b)
			}
		}
		// the token list carries its single bytes itself: none of the include* switches, nothing excluded (go :2742-2766)
		vocab, err := tokenmonster.NewVocab(tokens, special, c.Charset, c.Normalize, c.Capcode, false, false, false, false, false, false)
		must(err)
		tmp := filepath.Join(os.TempDir(), "go_oracle_"+c.Name+".vocab")
		must(vocab.Save(tmp))
		img, err := os.ReadFile(tmp)
		must(err)
		os.Remove(tmp)
		out := caseOut{
			Note:      "written by the reference's Go implementation (NewVocab, Save, Tokenize, Count)",
			Generator: "go_oracle/dump.go",
			Name:      c.Name, TokensB64: c.TokensB64, Special: c.Special, Capcode: c.Capcode, Charset: c.Charset, Normalize: c.Normalize,
			VocabB64: base64.StdEncoding.EncodeToString(img), DocsB64: c.DocsB64,
		}
		for _, d := range c.DocsB64 {
			doc, err := base64.StdEncoding.DecodeString(d)
			must(err)
			ids, missing, err := vocab.Tokenize(doc)
			must(err)
			n, _, err := vocab.Count(doc)
			must(err)
			if ids == nil {
				ids = []uint32{}
			}
			out.Ids = append(out.Ids, ids)
			out.Missing = append(out.Missing, missing)
			out.Count = append(out.Count, n)
		}
		js, err := json.Marshal(out)
		must(err)
		must(os.WriteFile(filepath.Join(os.Args[2], "go_"+c.Name+".json"), js, 0o644))
		fmt.Println("wrote go_" + c.Name + ".json")
	}
}
