#!/usr/bin/env Rscript
# tools/make_r_fixtures.R -- FOR A USER WHO HAS R AND THE REAL bWGR PACKAGE (neither exists in the build image of this repository).
#
# Runs the reference itself -- bWGR::KMUP, BayesA ... BayesDpi, wgr -- on its own data set (tpod) under set.seed(), and writes inputs' recipes and
# outputs as plain text under tests/golden/r_fixtures/.  tests/test_r_fixtures.py then runs this repository's CPU oracle in its R-stream mode
# (oracle/bwgr_rstream.h: Mersenne-Twister, inversion normals, nmath's rgamma / rbinom, restated and UNVERIFIED) with the same seed and compares:
# that is the only way the oracle -- and through it every GPU parity test -- can be pinned to real bWGR.  Nothing of the reference is copied here;
# the script only calls its exported functions.
#
#   Rscript tools/make_r_fixtures.R            (from the repository root; needs install.packages("bWGR"))
#
# Cases (kept equal to tests/golden/make_oracle_goldens.py where the reference allows):
#   kmup_pi0 / kmup_pi03 : KMUP(X, b, d, xx, e, L, Ve = 0.03, pi) under set.seed(77), inputs built without random numbers (recipe below)
#   <sampler>            : BayesA / B / C / L / RR / Cpi / Dpi (y, gen, it = 20, bi = 5, [pi = 0.9,] df = 5, R2 = 0.5) under set.seed(11)
#   wgr_<setting>        : wgr(y, gen, it = 25, bi = 5, ...) in the settings of man/wgr.Rd under set.seed(21)
suppressMessages(library(bWGR))
data(tpod)
out <- file.path("tests", "golden", "r_fixtures")
dir.create(out, recursive = TRUE, showWarnings = FALSE)
put <- function(case, name, x) {
  d <- file.path(out, case); dir.create(d, showWarnings = FALSE)
  writeLines(formatC(as.numeric(x), digits = 17, format = "g"), file.path(d, paste0(name, ".txt")))
}
n <- nrow(gen); p <- ncol(gen)
writeLines(c(paste("R", R.version.string), paste("bWGR", as.character(packageVersion("bWGR"))), paste("RNGkind", paste(RNGkind(), collapse = " / "))),
           file.path(out, "VERSIONS.txt"))

# ---- KMUP: deterministic inputs (no RNG), so that the Python side rebuilds them exactly
j <- seq_len(p)
b0 <- 0.01 * sin(j); d0 <- rep(1, p); xx <- colSums(gen^2)
e0 <- as.numeric(y - mean(y) - gen %*% b0)
L <- 120 * (1 + 0.5 * cos(j))
for (cs in list(list("kmup_pi0", 0), list("kmup_pi03", 0.3))) {
  set.seed(77)
  r <- KMUP(gen, b0, d0, xx, e0, L, 0.03, cs[[2]])
  put(cs[[1]], "b", r$b); put(cs[[1]], "d", r$d); put(cs[[1]], "e", r$e)
}

# ---- the fused samplers
smp <- list(BayesA = function() BayesA(y, gen, it = 20, bi = 5, df = 5, R2 = 0.5), BayesB = function() BayesB(y, gen, it = 20, bi = 5, pi = 0.9, df = 5, R2 = 0.5),
            BayesC = function() BayesC(y, gen, it = 20, bi = 5, pi = 0.9, df = 5, R2 = 0.5), BayesL = function() BayesL(y, gen, it = 20, bi = 5, df = 5, R2 = 0.5),
            BayesRR = function() BayesRR(y, gen, it = 20, bi = 5, df = 5, R2 = 0.5), BayesCpi = function() BayesCpi(y, gen, it = 20, bi = 5, df = 5, R2 = 0.5),
            BayesDpi = function() BayesDpi(y, gen, it = 20, bi = 5, df = 5, R2 = 0.5))
for (nm in names(smp)) {
  set.seed(11)
  r <- smp[[nm]]()
  for (f in names(r)) put(nm, f, r[[f]])
}

# ---- wgr()
wg <- list(BRR = list(), BayesA = list(iv = TRUE), BayesB = list(iv = TRUE, pi = 0.5), BayesC = list(pi = 0.5), BayesL = list(de = TRUE), thin = list(th = 3, bi = 4))
for (nm in names(wg)) {
  a <- modifyList(list(y = y, X = gen, it = 25, bi = 5), wg[[nm]])
  set.seed(21)
  r <- do.call(wgr, a)
  for (f in names(r)) put(paste0("wgr_", nm), f, r[[f]])
}
cat("wrote", out, "\n")
