# rshim/bwgr_hip.R -- R front-end over rshim/bwgr_shim.c: same names, argument order, defaults and return lists as
# R/RcppExports.R:4-6,48-74 and R/wgr.R:2-8, so that sourcing this file after library(bWGR) switches the Gibbs hot path to
# the MI355X engine.  SOURCE ONLY (no R in the build image; see INTEGRATION.md).
# dyn.load("bwgrhip.so")

.bwgr_panel <- function(X, device = 0L) if (inherits(X, "externalptr")) X else .Call("bwgrhip_panel", X, as.integer(device))
.bwgr_iter <- local({ i <- -1L; function() { i <<- i + 1L; i } })   # iteration word of the RNG counter for bare KMUP calls

KMUP <- function(X, b, d, xx, e, L, Ve, pi) .Call("bwgrhip_KMUP", .bwgr_panel(X), as.double(b), as.double(d), as.double(xx), as.double(e), as.double(L), as.double(Ve), as.double(pi), .bwgr_iter())
KMUP2 <- function(X, Use, b, d, xx, E, L, Ve, pi) .Call("bwgrhip_KMUP2", .bwgr_panel(X), as.double(Use), as.double(b), as.double(d), as.double(xx), as.double(E), as.double(L), as.double(Ve), as.double(pi), .bwgr_iter())

.bwgr_fused <- function(model, y, X, it, bi, pi, df, R2) .Call("bwgrhip_Bayes", as.integer(model), as.double(y), .bwgr_panel(X), as.double(it), as.double(bi), as.double(pi), as.double(df), as.double(R2))
BayesA   <- function(y, X, it = 1500, bi = 500, df = 5, R2 = 0.5) .bwgr_fused(0L, y, X, it, bi, 0, df, R2)
BayesB   <- function(y, X, it = 1500, bi = 500, pi = 0.95, df = 5, R2 = 0.5) .bwgr_fused(1L, y, X, it, bi, pi, df, R2)
BayesC   <- function(y, X, it = 1500, bi = 500, pi = 0.95, df = 5, R2 = 0.5) .bwgr_fused(2L, y, X, it, bi, pi, df, R2)
BayesL   <- function(y, X, it = 1500, bi = 500, df = 5, R2 = 0.5) .bwgr_fused(3L, y, X, it, bi, 0, df, R2)
BayesRR  <- function(y, X, it = 1500, bi = 500, df = 5, R2 = 0.5) .bwgr_fused(4L, y, X, it, bi, 0, df, R2)
BayesCpi <- function(y, X, it = 1500, bi = 500, df = 5, R2 = 0.5) .bwgr_fused(5L, y, X, it, bi, 0, df, R2)
BayesDpi <- function(y, X, it = 1500, bi = 500, df = 5, R2 = 0.5) .bwgr_fused(6L, y, X, it, bi, 0, df, R2)
# two-effect samplers, R/RcppExports.R (BayesA2, BayesB2, BayesRR2): the two panels must share the slab geometry, so the
# second is staged with the first one's workgroup count
.bwgr_fused2 <- function(model, y, X1, X2, it, bi, pi, df, R2) .Call("bwgrhip_Bayes2", as.integer(model), as.double(y), .bwgr_panel(X1), .bwgr_panel(X2), as.double(it), as.double(bi), as.double(pi), as.double(df), as.double(R2))
BayesA2  <- function(y, X1, X2, it = 1500, bi = 500, df = 5, R2 = 0.5) .bwgr_fused2(0L, y, X1, X2, it, bi, 0, df, R2)
BayesB2  <- function(y, X1, X2, it = 1500, bi = 500, pi = 0.95, df = 5, R2 = 0.5) .bwgr_fused2(1L, y, X1, X2, it, bi, pi, df, R2)
BayesRR2 <- function(y, X1, X2, it = 1500, bi = 500, df = 5, R2 = 0.5) .bwgr_fused2(4L, y, X1, X2, it, bi, 0, df, R2)

wgr <- function(y, X, it = 1500, bi = 500, th = 1, bag = 1, rp = FALSE, iv = FALSE, de = FALSE, pi = 0, df = 5, R2 = 0.5,
                eigK = NULL, VarK = 0.95, verb = FALSE) {
  if (bag != 1 && !is.null(eigK)) stop("bag != 1 with eigK is undefined in bWGR (R/wgr.R:73-79); not supported")
  if (bag != 1) df <- df   # df/(bag^2) (R/wgr.R:20) is applied inside the engine
  if (anyNA(X)) {                       # R/wgr.R:12-18
    imp <- function(x) { x[is.na(x)] <- mean(x, na.rm = TRUE); x[is.nan(x)] <- 0; x }
    X <- apply(X, 2, imp)
  }
  gen0 <- X; mis <- integer(0)           # R/wgr.R:10: predictions are returned for every row of gen0
  if (anyNA(y)) { mis <- which(is.na(y)); y <- y[-mis]; X <- X[-mis, , drop = FALSE] }   # R/wgr.R:34-39
  U <- U0 <- V <- NULL
  if (!is.null(eigK)) {                  # R/wgr.R:23-27; the rows of missing y are dropped from U for the sweeps (:37)
    V <- eigK$values; pk <- which.max((cumsum(V) / length(V)) > VarK)
    U0 <- eigK$vectors[, 1:pk, drop = FALSE]; V <- V[1:pk]
    U <- if (length(mis)) U0[-mis, , drop = FALSE] else U0
  }
  fit <- .Call("bwgrhip_wgr", as.double(y), .bwgr_panel(X), as.integer(it), as.integer(bi), as.integer(th), as.logical(iv), as.logical(de),
               as.double(pi), as.double(df), as.double(R2), U, V, as.double(bag), as.logical(rp))
  if (length(mis)) {                     # R/wgr.R:146-153: HAT = B0 + gen0 %*% B (+ U0 %*% H) over ALL rows, missing-y rows included
    hat <- matrix(0, nrow(gen0), 1); hat[-mis, 1] <- fit$hat
    hat[mis, 1] <- fit$mu + gen0[mis, , drop = FALSE] %*% fit$b
    if (!is.null(U0)) {
      H <- qr.solve(U, fit$u)            # u = U %*% H on the swept rows; H recovered for the rows that were left out
      poly <- U0 %*% H; hat[mis, 1] <- hat[mis, 1] + poly[mis]; fit$u <- poly
    }
    fit$hat <- hat
  }
  fit
}

# EM / Gauss-Seidel family, R/RcppExports.R (emRR, emBA, emBB, emBC, emBCpi, emDE, emBL, emEN, emML): same names, argument
# order and defaults; the marker order of every sweep is the reference's std::shuffle(order, std::mt19937(i))
.bwgr_em <- function(model, y, gen, df, R2, par, D = NULL) .Call("bwgrhip_em", as.integer(model), as.double(y), .bwgr_panel(gen), as.double(df), as.double(R2), as.double(par), D)
emRR   <- function(y, gen, df = 10, R2 = 0.5) .bwgr_em(0L, y, gen, df, R2, 0)
emBA   <- function(y, gen, df = 10, R2 = 0.5) .bwgr_em(1L, y, gen, df, R2, 0)
emDE   <- function(y, gen, R2 = 0.5) .bwgr_em(2L, y, gen, 0, R2, 0)
emML   <- function(y, gen, D = NULL) .bwgr_em(3L, y, gen, 0, 0.5, 0, D)
emBB   <- function(y, gen, df = 10, R2 = 0.5, Pi = 0.75) .bwgr_em(4L, y, gen, df, R2, Pi)
emBC   <- function(y, gen, df = 10, R2 = 0.5, Pi = 0.75) .bwgr_em(5L, y, gen, df, R2, Pi)
emBCpi <- function(y, gen, df = 10, R2 = 0.5, Pi = 0.75) .bwgr_em(6L, y, gen, df, R2, Pi)
emBL   <- function(y, gen, R2 = 0.5, alpha = 0.02) .bwgr_em(7L, y, gen, 0, R2, alpha)
emEN   <- function(y, gen, R2 = 0.5, alpha = 0.02) .bwgr_em(8L, y, gen, 0, R2, alpha)
lasso  <- function(y, gen) .bwgr_em(9L, y, gen, 0, 0.5, 0)
