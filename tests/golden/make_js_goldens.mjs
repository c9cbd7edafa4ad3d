// Step 2 of the JS-tier goldens: run every case of tests/golden/js/cases.json through the REFERENCE's own
// JavaScript CPU implementation of the DP path (js/webgpu/cpu/{forward,backward,viterbi}-2d.mjs: Float64,
// exact logsumexp, dense (nIn,nOut,S,S) transition tensor), imported in place from /root/reference, and store
// the outputs as data in tests/golden/js/goldens.json.  Only this container has /root/reference; the GPU box and
// the tests read the committed JSON.
//
//   node tests/golden/make_js_goldens.mjs        (Node >= 12.17)
import { readFileSync, writeFileSync } from 'fs';
import { dirname, join } from 'path';
import { fileURLToPath } from 'url';
import { prepareMachine } from '/root/reference/js/webgpu/internal/machine-prep.mjs';
import { forward2DFull } from '/root/reference/js/webgpu/cpu/forward-2d.mjs';
import { backward2D } from '/root/reference/js/webgpu/cpu/backward-2d.mjs';
import { viterbi2D } from '/root/reference/js/webgpu/cpu/viterbi-2d.mjs';
// ... and its 1-D tier for generators / recognisers (one tape; BASELINE config 5's family): cases with `oneTape`
import { forward1DFull } from '/root/reference/js/webgpu/cpu/forward-1d.mjs';
import { backward1D } from '/root/reference/js/webgpu/cpu/backward-1d.mjs';
import { viterbi1D } from '/root/reference/js/webgpu/cpu/viterbi-1d.mjs';

const here = dirname(fileURLToPath(import.meta.url));
const cases = JSON.parse(readFileSync(join(here, 'js', 'cases.json'), 'utf8'));
const num = (x) => (x === -Infinity ? '-inf' : (x === Infinity ? 'inf' : x));

async function main() {
  const out = [];
  for (const c of cases) {
    const mj = JSON.parse(readFileSync(join(here, c.machine), 'utf8'));
    const pm = prepareMachine(mj, c.params);
    if (pm.nStates !== c.nStates) throw new Error('state count mismatch for ' + c.name);
    const x = Uint32Array.from(c.input), y = Uint32Array.from(c.output);
    const Li = x.length, Lo = y.length, S = pm.nStates;
    if (c.oneTape) {
      // the 1-D tier: null for the tape the machine does not have; grids are [(position)*nStates + state], position 0..L
      const xi = c.oneTape === 'in' ? x : null, yo = c.oneTape === 'in' ? null : y;
      if ((c.oneTape === 'in' ? pm.nOutputTokens : pm.nInputTokens) !== 1) throw new Error('not a one-tape machine: ' + c.name);
      const f = await forward1DFull(pm, xi, yo);
      const b = await backward1D(pm, xi, yo);
      const v = await viterbi1D(pm, xi, yo);
      const rec = { name: c.name, forward: num(f.logLikelihood), backward: num(b.logLikelihood), viterbi: num(v.score), layout: 'cells[p*nStates+s]', oneTape: c.oneTape };
      const n = (Math.max(Li, Lo) + 1) * S;
      if (n <= 6000) { rec.forwardCells = Array.from(f.dp, num); rec.backwardCells = Array.from(b.bp, num); }
      else { rec.sample = []; for (let k = 0; k < n; k += 89) rec.sample.push([k, num(f.dp[k]), num(b.bp[k])]); }
      out.push(rec);
      console.log(c.name, f.logLikelihood, b.logLikelihood, v.score);
      continue;
    }
    const f = await forward2DFull(pm, x, y);
    const b = await backward2D(pm, x, y);
    const v = await viterbi2D(pm, x, y);
    const rec = { name: c.name, forward: num(f.logLikelihood), backward: num(b.logLikelihood), viterbi: num(v.score),
                  layout: 'cells[(i*(outLen+1)+o)*nStates+s]' };
    // full matrices for the small machines; for wide ones a deterministic sample of cells
    const n = (Li + 1) * (Lo + 1) * S;
    if (n <= 4000) {
      rec.forwardCells = Array.from(f.dp, num);
      rec.backwardCells = Array.from(b.bp, num);
    } else {
      rec.sample = [];
      for (let k = 0; k < n; k += 97) rec.sample.push([k, num(f.dp[k]), num(b.bp[k])]);
    }
    out.push(rec);
    console.log(c.name, f.logLikelihood, b.logLikelihood, v.score);
  }
  writeFileSync(join(here, 'js', 'goldens.json'), JSON.stringify(out));
}
main();
