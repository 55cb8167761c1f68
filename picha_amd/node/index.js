"use strict";
// picha (pixel hot path) on MI355X: the same four calls and the same Image type as the reference's
// index.js:13-33, backed by the HIP kernels in libpicha_hip.so through ./build/Release/picha.node.
// Codecs are out of scope: `catalog` is empty, so code that feature-tests codecs (as the reference's
// own tests do) sees none.

const native = require('./build/Release/picha.node');
const Image = exports.Image = require('./lib/image').Image;

exports.catalog = native.catalog;

function wrap(cb) {
	return function (err, dst) { return cb(err, dst && new Image(dst)); };
}

exports.resize = function (src, opt, cb) { native.resize(src, opt, wrap(cb)); };
exports.resizeSync = function (img, opt) { return new Image(native.resizeSync(img, opt)); };
exports.colorConvert = function (src, opt, cb) { native.colorConvert(src, opt, wrap(cb)); };
exports.colorConvertSync = function (img, opt) { return new Image(native.colorConvertSync(img, opt)); };

// outside the drop-in surface
// resizeSync(colorConvertSync(img, opt), opt) — or the other order with opt.order === 'resizeFirst' — in one device
// round trip: the intermediate image never leaves the GPU; same result as the two calls
exports.convertResize = function (src, opt, cb) { native.convertResize(src, opt, wrap(cb)); };
exports.convertResizeSync = function (img, opt) { return new Image(native.convertResizeSync(img, opt)); };
exports.hipDeviceCount = native.hipDeviceCount;
exports.hipLastKernel = native.hipLastKernel;
